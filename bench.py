#!/usr/bin/env python
"""Headline benchmark: frames/sec of the Context-Aware PoseFormer hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU over RCCL.  Either the driver launches the ranks (`python -m torch.distributed.run
--nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE come from the environment) or,
when WORLD_SIZE is not set, this script re-executes itself under torch.distributed.run with N ranks on 127.0.0.1.

A "step" is ONE pass of the hot path (CA_PF.forward through the C ABI: backbone -> joint-context sampling -> lifting
transformer; with --train also MPJPE, the lifter backward, the flat-gradient all-reduce and fused AdamW) over one
batch of synthetic input already resident in HBM.  Default workload = BASELINE.json configs[1]: batch 64 per GPU,
HRNet-32, 256x256, fp32.  The other configurations are selected with --config 0..4 (or spelled out with --backbone /
--batch / --height / --width / --dtype / --train); frames are independent, so N ranks share nothing on the data path
(weak scaling, no collective) except the one gradient all-reduce of the training configuration.

Rank 0 prints ONE JSON line: whole-job frames/s plus
  roofline     — for the dominant kernel: algorithmic FLOPs (and algorithmic HBM bytes) of its launches / their summed
                 duration, measured with HIP event pairs on the launch stream (capf_forward_profile_launches), against
                 the dense MFMA peak for the dtype and the 8 TB/s HBM peak (MI355X_MICROARCH.md); `bound` names the
                 roof the kernel sits closer to, `other_roof` carries the second one; `traffic` = measured HBM bytes
                 per launch from the offline rocprofv3 FETCH_SIZE / WRITE_SIZE passes of the same command
                 (profiles/r04_hbm_traffic.json, keyed by configuration), null if not collected;
  cpu_baseline — the CPU oracle (a port of the reference forward to functional PyTorch-CPU) timed on this box's host
                 cores on bounded samples of the same workload (rank 0, N=1 only): batch 1 and batch 16;
  vs_fp32_oracle — accuracy beside the speed (same leg as cpu_baseline, inference): max |joint coordinate difference| and mean
                 per-joint distance (metres) between the timed step's output and the fp32 CPU oracle on four of the run's own
                 frames; --lifter-fp32 (plan flag CAPF_PLAN_LIFTER_FP32) keeps the lifter's projections fp32 under --dtype bf16.
fp32 configurations: the 3x3 stride-1 convs run on the bf16 matrix pipe with every fp32 operand split EXACTLY into three bf16 numbers
(csrc/igemm_f32x3_ws.hip: results as close to fp64 as the direct fp32 MFMA kernel's); --no-f32x3 (CAPF_PLAN_NO_F32X3) times
round 3's plan (Winograd on the fp32 pipe) for comparison, and every fp32 inference line carries that plan's rate for the same K steps as
`fp32_pipe_plan` (never `value`).  `dtype` stays "f32": that is the arithmetic the path computes in.
"""
import argparse
import copy
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "contextaware-poseformer_amd"))

PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}   # MI355X_MICROARCH.md: dense MFMA peaks (fp32-input matrix = vector rate)
HBM_PEAK_GBS = 8000.0

# the reference's own forward timed in the survey container (not on this box; orientation only)
REFERENCE_CPU = {
    ("hrnet_32", 256, 256): {"cores": 8, "kind": "reference", "where": "survey container, 8 x Xeon 2.10 GHz",
                             "frames_per_s": {"batch 1": 9.0, "batch 16": 18.8, "batch 64": 13.9}},
    ("hrnet_48", 256, 256): {"cores": 8, "kind": "reference", "where": "survey container, 8 x Xeon 2.10 GHz",
                             "frames_per_s": {"batch 16": 13.3}},
    ("cpn", 384, 288): {"cores": 8, "kind": "reference", "where": "survey container, 8 x Xeon 2.10 GHz",
                        "frames_per_s": {"batch 16": 9.3}},
}

# BASELINE.json configs[i] -> arguments (batch is per GPU)
CONFIGS = {
    0: dict(backbone="hrnet_32", batch=1, height=256, width=256, dtype="f32", train=False),
    1: dict(backbone="hrnet_32", batch=64, height=256, width=256, dtype="f32", train=False),
    2: dict(backbone="hrnet_48", batch=256, height=256, width=256, dtype="bf16", train=False),
    3: dict(backbone="hrnet_32", batch=512, height=256, width=256, dtype="f32", train=True),
    4: dict(backbone="cpn", batch=128, height=384, width=288, dtype="bf16", train=False),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=None, choices=sorted(CONFIGS), help="BASELINE.json configs[i]; explicit flags override")
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (default 64; 512 with --train)")
    ap.add_argument("--train", action="store_true", default=None, help="configs[3]: training step (frozen backbone forward, "
                    "lifter forward+backward, MPJPE, gradient all-reduce, fused AdamW) instead of inference")
    ap.add_argument("--backbone", default=None, choices=["hrnet_32", "hrnet_48", "cpn"])
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16"], help="bf16: backbone convs + lifter GEMM operands on bf16 MFMA")
    ap.add_argument("--embed", type=int, default=128, help="poseformer.embed_dim_ratio (128 = the reference default; 256 = the "
                    "labelled extra point for BASELINE's 'dim=256')")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-plan", action="store_true", help="skip the extra measurement of the fp32-pipe plan (`fp32_pipe_plan` in the line)")
    ap.add_argument("--no-f32x3", action="store_true", help="fp32 runs: 3x3 convs on the Winograd kernels at every batch (CAPF_PLAN_NO_F32X3: round 3's plan) instead of the split-fp32 tile")
    ap.add_argument("--x3-exact", action="store_true", help="fp32 runs: round 4's exact three-bf16-piece tile (six piece products, CAPF_PLAN_F32X3_EXACT) instead of the "
                    "two-fp16-piece tile (three)")
    ap.add_argument("--lifter-fp32", action="store_true", help="bf16 runs: lifter projections on the fp32 kernels (CAPF_PLAN_LIFTER_FP32)")
    ap.add_argument("--plan-flags", type=int, default=0, help="extra capf_plan_flag bits OR-ed into the plan (A/B runs of one kernel family; named in config.plan_flags)")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--kernel-table", action="store_true", help="print the per-kernel table to stderr")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo for smoke tests)")
    ap.add_argument("--lanes", type=int, default=-1,
                    help="independent backbone branches: 3 two grouped chains on two streams (default), 2 one grouped chain, 1 side streams, "
                         "0 program order")
    ap.add_argument("--overlap", type=int, default=2,
                    help="inference, rank 0 at N=1: after the contract's one-batch-at-a-time measurement, ALSO time the same K steps "
                         "issued round-robin over this many engines (own workspaces, own HIP streams) so that consecutive batches "
                         "overlap; reported under \"overlapped_steps\", never as \"value\" (0 / 1 = skip)")
    ap.add_argument("--sustain", type=float, default=2.5,
                    help="rank 0 at N=1: after the timed region, loop the same step for at least this many seconds with board power and "
                         "shader clock sampled (rocm-smi) -- reported under \"sustained\", never as \"value\" (0 = skip)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak (default, the driver's contract) = --batch frames on EVERY GPU; strong (BASELINE.md section 4 item 4) = --batch is the "
                         "GLOBAL batch, split evenly across the ranks (shard_bounds), so total work is fixed as N grows")
    ap.add_argument("--dry-run", action="store_true", help="rendezvous, sharding, barriers, max-over-ranks and the JSON line "
                    "with a sleep instead of the GPU step (CPU smoke test of the N>1 flow)")
    a = ap.parse_args(argv)
    base = dict(CONFIGS[a.config if a.config is not None else 1])
    if a.train and a.config is None:
        base.update(batch=512, train=True)
    for k in ("backbone", "batch", "height", "width", "dtype", "train"):
        if getattr(a, k) is None:
            setattr(a, k, base[k])
    return a


def config_tag(a):
    """Which BASELINE.json configuration the arguments spell, if any."""
    cur = dict(backbone=a.backbone, batch=a.batch, height=a.height, width=a.width, dtype=a.dtype, train=bool(a.train))
    for i, c in CONFIGS.items():
        if c == cur and a.embed == 128:
            return i
    return None


def workload_string(a, tag):
    head = f"configs[{tag}]" if tag is not None else "custom (not a BASELINE.json configuration)"
    per = (f"batch {a.batch}/GPU" if getattr(a, "scaling", "weak") == "weak" or a.gpus == 1 else
           f"GLOBAL batch {a.batch} split over {a.gpus} GPUs ({a.batch // a.gpus}/GPU, strong scaling)")
    arith = (("fp32 tensors, fp32 accumulation; 3x3 stride-1 convs from batch 5: " +
              ("fp32 matrix pipe (Winograd / direct; CAPF_PLAN_NO_F32X3)" if getattr(a, "no_f32x3", False) else
               "each operand split exactly into three bf16 pieces, six piece products on the bf16 matrix pipe (CAPF_PLAN_F32X3_EXACT)" if getattr(a, "x3_exact", False) else
               "each operand as two fp16 pieces under exact power-of-two block scales (to 2^-23), three piece products on the 16-bit matrix pipe -- same "
               "measured distance to fp64 as the direct fp32 MFMA kernel; Inf / NaN / |x| >= 1.5e23 give NaN (igemm_f32h2_ws); the exact-operand and "
               "fp32-pipe plans are timed on the same line (exact_split_plan, fp32_pipe_plan)") +
              ("; every other conv / GEMM: fp32 MFMA" if getattr(a, "no_f32x3", False) else
               "; the other convs and the lifter's projections (LayerNorm-folded ones included): the same two-piece arithmetic (igemm_f32h2g); "
               "pointwise layer1 convs, the stem and the lifter's embed_proj / feat_embed rows: fp32 MFMA")) if a.dtype == "f32" else
             "bf16 MFMA operands, fp32 accumulate (backbone convs; lifter GEMMs fp32: --lifter-fp32)" if getattr(a, "lifter_fp32", False) else
             "bf16 MFMA operands, fp32 accumulate (backbone convs, layer1's first bottleneck as one kernel, lifter GEMMs; LN / softmax / samplers / "
             "embed_proj / feat_embed / residual stream fp32)")
    if a.train:
        return (f"{head}: TRAINING step, {per} {a.backbone} {a.height}x{a.width} (frozen backbone forward, lifter "
                f"fwd+bwd, MPJPE, flat-gradient all-reduce, fused AdamW, DropPath on), lifter embed {a.embed} levels 4, {arith}")
    return (f"{head}: {per} {a.backbone} {a.height}x{a.width} image + 17 kpts -> 17x3, PoseFormer lifter embed "
            f"{a.embed} levels 4, {arith}, inference")


def respawn_under_torchrun(a):
    """--gpus N without a launcher: run N ranks of this same command under torch.distributed.run on 127.0.0.1."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def vs_fp32_oracle(backbone, sd_cpu, img, k2d, kc, got, others=None):
    """The timed step's output against the fp32 CPU oracle on four of the run's own frames (the oracle is the checker here, as in
    smoke(); part of the cpu_baseline leg).  `others`: outputs of the same step on the alternative plans, held to the same frames."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import capf_oracle as oracle
    B = img.shape[0]
    idx = sorted({0, B // 3, (2 * B) // 3, B - 1})
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        want = oracle.ca_pf_forward(sd_cpu, img[idx].cpu(), k2d[idx].cpu(), kc[idx].cpu().clone(), backbone=backbone)
    d = got[idx].cpu().float() - want
    res = {"frames": idx, "max_abs": float(d.abs().max()), "mean_joint_dist": float(d.norm(dim=-1).mean()), "unit": "m",
           "reference": "oracle/capf_oracle.py, fp32 (pinned to the reference's outputs, tests/golden)"}
    for key, o in (others or {}).items():
        dd = o[idx].cpu().float() - want
        res[key] = {"max_abs": float(dd.abs().max()), "mean_joint_dist": float(dd.norm(dim=-1).mean())}
    return res


def reference_in_bf16(backbone):
    """The yardstick for a bf16 line's `vs_fp32_oracle`: how far the REAL reference moves from its own fp32 joints when it is evaluated in
    bf16 (committed fixture tests/golden/bf16_reference.npz, made by oracle/make_goldens.py from the imported reference on a golden frame --
    other frames and weights than this run's, same architecture and input size)."""
    import numpy as np
    case = {"hrnet_48": "w48_256x256_b1", "cpn": "cpn_384x288_b1"}.get(backbone)
    path = os.path.join(ROOT, "tests", "golden", "bf16_reference.npz")
    if case is None or not os.path.exists(path):
        return None
    d = np.load(path, allow_pickle=False)
    get = lambda tag: {"max_abs": float(d[f"{case}:{tag}:joints_maxabs"]), "mean_joint_dist": float(d[f"{case}:{tag}:joints_mean_dist"])}
    return {"golden_case": case, "unit": "m", "torch_autocast_cpu_bfloat16": get("ac"), "fp32_activations_bf16_conv_linear_operands": get("opr"),
            "note": "|reference in bf16 - reference in fp32| on the golden frame: the reference's own bf16 noise floor"}


def cpu_baseline(backbone, H, W, sd_cpu, budget_s=24.0):
    """Time the CPU oracle on bounded samples of the workload: batch 1 (latency) and batch 16 (throughput), as SURVEY.md
    §8(d) plans.  oneDNN convolutions stop scaling (and can collapse) far below the core count of a 256-core host, so
    for batch 16 the thread count is swept 16/32/64/all while it keeps paying; `cores` is the thread count of the reported
    number, `host_cores` what the box has.  `value` is the best batch-16 rate."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import capf_oracle as oracle
    from capf import synth
    host = os.cpu_count() or 1
    t_start = time.perf_counter()

    def run(B, threads, reps):
        img, k2d, kc = synth.synth_inputs(B, H, W, seed=101)
        torch.set_num_threads(threads)
        ts = []
        with torch.no_grad():
            for _ in range(reps):
                t0 = time.perf_counter()
                oracle.ca_pf_forward(sd_cpu, img, k2d, kc.clone(), backbone=backbone)
                ts.append(time.perf_counter() - t0)
        return ts

    points = []
    best_threads, best = None, None
    for threads in sorted({t for t in (16, 32, 64, host) if t <= host}):
        t = min(run(16, threads, 2))                      # first call doubles as warm-up
        points.append({"batch": 16, "threads": threads, "frames_per_s": round(16 / t, 2)})
        if best is None or t < best:
            best_threads, best = threads, t
        if t > 1.25 * best or time.perf_counter() - t_start > budget_s * 0.5:
            break
    times = run(16, best_threads, 1)
    while time.perf_counter() - t_start < budget_s * 0.75 and len(times) < 20:
        times += run(16, best_threads, 1)
    times.sort()
    med16 = times[len(times) // 2]
    t1 = run(1, min(best_threads, host), 4)[1:]
    t1.sort()
    points.append({"batch": 1, "threads": best_threads, "frames_per_s": round(1 / t1[len(t1) // 2], 2)})
    return {"value": round(16 / med16, 2), "unit": "frames/s", "cores": best_threads, "host_cores": host, "kind": "port",
            "points": points,
            # the REFERENCE itself (imported from /root/reference, PyTorch-CPU/oneDNN) cannot travel to this box; its timing in the
            # survey container is the second comparator SURVEY.md §8(d) asks for (8 x Xeon 2.1 GHz, 8 threads)
            "reference_in_survey_container": REFERENCE_CPU.get((backbone, H, W)),
            "sample": f"{len(times)} x batch-16 and {len(t1)} x batch-1 {backbone} {H}x{W} fp32 forwards of oracle/capf_oracle.py "
                      f"(functional PyTorch-CPU/oneDNN; value = median batch-16 rate at {best_threads} threads, the best of the sweep in `points`)"}


def input_seed(rank):
    """Every rank draws its OWN shard of synthetic frames (data parallel: different frames per GPU)."""
    return 1000 + rank


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_torchrun(a))
    import torch
    import torch.distributed as dist
    from capf import dist as cdist
    rank, world, local = cdist.init_from_env(a.backend)    # "nccl" is RCCL on ROCm; no-op when WORLD_SIZE == 1
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    tag = config_tag(a)
    B, H, W = a.batch, a.height, a.width
    global_frames = B * world
    if a.scaling == "strong":                              # the global batch is fixed: this rank's shard of it (frames are independent)
        if B % world != 0:
            raise SystemExit(f"--scaling strong: the global batch {B} does not divide over {world} ranks")
        global_frames = B
        lo_s, hi_s = cdist.shard_bounds(B, rank, world)
        B = hi_s - lo_s

    if a.dry_run:
        lo, hi = cdist.shard_bounds(global_frames, rank, world)
        cdist.barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            time.sleep(0.002 * (1 + rank))
        own_s = time.perf_counter() - t0                       # this rank's own time (the per-rank rate) ...
        cdist.barrier()
        mine_s = time.perf_counter() - t0                      # ... and the bracketed one (max over ranks -> value)
        dry_dist = None
        if world > 1:
            ones = torch.ones(1)
            dist.all_reduce(ones)
            mine = torch.tensor([B * a.steps / own_s], dtype=torch.float64)
            rates = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(rates, mine)
            seed = torch.tensor([input_seed(rank)], dtype=torch.int64)
            seeds = [torch.empty_like(seed) for _ in range(world)]
            dist.all_gather(seeds, seed)
            dry_dist = {"backend": dist.get_backend(), "ranks_seen": int(ones.item()), "devices_visible": torch.cuda.device_count(),
                        "per_rank_frames_per_s": [round(r.item(), 2) for r in rates], "input_seeds": [int(x.item()) for x in seeds]}
        elapsed = cdist.max_over_ranks(mine_s, torch.device("cpu"))
        if rank == 0:
            print(json.dumps({"metric": "frames/sec", "value": round(global_frames * a.steps / elapsed, 2), "unit": "frames/s",
                              "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 4),
                              "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": a.dtype, "data": "none",
                              "dry_run": True, "config": {"workload": workload_string(a, tag), "frames_per_step": global_frames,
                                                          "frames_per_gpu": B, "shard_of_rank0": [lo, hi]},
                              "distributed": dry_dist}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    from capf import synth
    from mvn.models.conpose import CA_PF
    from mvn.utils.cfg import backbone_preset, config
    if os.environ.get("CAPF_BENCH_SINGLE_DEVICE"):         # smoke-testing the N>1 flow on a 1-GPU box (gloo only)
        local = 0
    elif world > 1 and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but this node shows {torch.cuda.device_count()} GPU(s): one rank per GPU is the contract "
                         f"(RCCL refuses two ranks on one device); nothing was measured")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    cfg = backbone_preset(copy.deepcopy(config), a.backbone)
    cfg.model.backbone.fix_weights = True
    cfg.model.poseformer.embed_dim_ratio = a.embed
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        from capf.lib import PLAN_F32X3_EXACT, PLAN_LIFTER_FP32, PLAN_NO_F32H2_GEMM, PLAN_NO_F32X3
        pflags = (PLAN_LIFTER_FP32 if (a.lifter_fp32 and a.dtype == "bf16") else 0) | a.plan_flags
        if a.no_f32x3:
            pflags |= PLAN_NO_F32X3 | PLAN_NO_F32H2_GEMM
        if a.x3_exact and a.dtype != "bf16":
            pflags |= PLAN_F32X3_EXACT
        model = CA_PF(cfg, compute_dtype="bf16" if a.dtype == "bf16" else "fp32", plan_flags=pflags).eval()
    sd_cpu = synth.load_synthetic(model, seed=1, bn_mode="random")
    model = model.to(dev)

    img, k2d, kc, gt = synth.synth_inputs(B, H, W, seed=input_seed(rank), crop_range=(192, 256), with_gt=True)
    img, k2d, kc0, gt = img.to(dev), k2d.to(dev), kc.to(dev), gt.to(dev)
    kc_work = kc0.clone()
    stream = torch.cuda.current_stream(dev)
    if a.lanes >= 0:
        model.engine_for(img).set_lanes(a.lanes)

    if a.train:
        from capf.optim import FusedAdamW, flatten_
        from mvn.models.loss import MPJPE
        model.train(); model.backbone.eval(); model.volume_net.train()      # train.py:144-148
        cdist.broadcast_state_(model.volume_net)                             # DDP ctor broadcast (C1)
        flat_p = flatten_(model.volume_net)
        opt = FusedAdamW(flat_p, lr=6.4e-4, weight_decay=0.1)                # train.py:345, human36m.yaml:58
        model.flat_grad_only = True                                          # gradient -> all-reduce -> AdamW on the flat buffer
        crit = MPJPE()

        phase_ev = None                                                       # (after the timed region: event pairs around the phases)

        def step():
            mark = (lambda: None) if phase_ev is None else (lambda: (phase_ev.append(torch.cuda.Event(enable_timing=True)), phase_ev[-1].record()))
            mark()
            kc_work.copy_(kc0)
            pred = model(img, k2d, kc_work)                                  # DropPath active (dpr 0..0.2)
            loss = crit(pred, gt)
            mark()
            model.zero_grad(set_to_none=True)
            loss.backward()
            mark()
            flat_g = model.last_flat_grad
            flat_g, gscale = cdist.allreduce_sum_(flat_g)                    # ONE RCCL all-reduce of 56.4 MB (C3) ...
            mark()
            opt.step(flat_g, grad_scale=gscale)                              # ... its 1 / world folded into the AdamW kernel
            model.lifter_params_changed()
            mark()
            return pred.detach()
    else:
        def step():
            kc_work.copy_(kc0)         # the forward normalises its 3rd argument in place (conpose.py:34-35)
            return model(img, k2d, kc_work)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with (torch.enable_grad() if a.train else torch.no_grad()):
        for _ in range(a.warmup):
            out = step()
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = step()
        torch.cuda.synchronize(dev)
        own_elapsed = time.perf_counter() - t0                 # this rank alone (per_rank_frames_per_s)
        fence()
        elapsed = time.perf_counter() - t0                     # bracketed by barrier + synchronize on both sides
    assert torch.isfinite(out).all()
    train_phases = None
    if a.train:                                            # untimed extra steps: where a training step's time goes (HIP events on the stream)
        acc_ms = [0.0, 0.0, 0.0, 0.0]
        nrep = 3
        with torch.enable_grad():
            for _ in range(nrep):
                phase_ev = []
                step()
                torch.cuda.synchronize(dev)
                for k in range(4):
                    acc_ms[k] += phase_ev[k].elapsed_time(phase_ev[k + 1])
        phase_ev = None
        train_phases = {"forward_loss_ms": round(acc_ms[0] / nrep, 3), "backward_ms": round(acc_ms[1] / nrep, 3),
                        "allreduce_ms": round(acc_ms[2] / nrep, 3), "optimizer_ms": round(acc_ms[3] / nrep, 3),
                        "note": "HIP events on the step's stream over 3 untimed extra steps; optimizer = fused AdamW + the lifter repack"}

    def measure_sustained():
        # ---- the SAME step looped for >= a.sustain seconds with board power and shader clock sampled from rocm-smi meanwhile: the timed
        # region above is K steps (0.1 - 1 s) on a part that sits on a 1.4 kW package cap -- this says what rate and clock the step
        # HOLDS.  Reported under "sustained", never as `value`.
        if world != 1 or a.sustain <= 0:
            return None
        import re
        import shutil
        import threading
        smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
        samples, stop = [], threading.Event()

        def sampler():
            while not stop.is_set():
                try:
                    txt = subprocess.run([smi, "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
                except Exception:
                    return
                pw = re.search(r"Power \(W\):\s*([0-9.]+)", txt)
                ck = re.search(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", txt)
                if pw and ck:
                    samples.append((time.perf_counter(), float(pw.group(1)), int(ck.group(1))))

        th = threading.Thread(target=sampler, daemon=True)
        n = 0
        with (torch.enable_grad() if a.train else torch.no_grad()):
            torch.cuda.synchronize(dev)
            th.start()
            t0 = time.perf_counter()
            while True:
                for _ in range(10):
                    step()
                n += 10
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                if t1 - t0 >= a.sustain:
                    break
        stop.set()
        th.join(timeout=15)
        under = sorted((p_, c_) for t_, p_, c_ in samples if t0 + 0.3 <= t_ <= t1)          # (samples taken under load)
        med = lambda v: v[len(v) // 2] if v else None
        pw, ck = sorted(x[0] for x in under), sorted(x[1] for x in under)
        return {"seconds": round(t1 - t0, 2), "steps": n, "frames_per_s": round(n * B / (t1 - t0), 2), "ms_per_step": round((t1 - t0) / n * 1e3, 4),
                "power_w": {"min": pw[0], "median": med(pw), "max": pw[-1]} if pw else None,
                "sclk_mhz": {"min": ck[0], "median": med(ck), "max": ck[-1]} if ck else None, "samples": len(under),
                "note": "the timed step looped back to back (host sync every 10 steps) with rocm-smi --showpower --showclocks sampled in a thread; "
                        "not `value`"}

    def measure_overlapped():
        # ---- consecutive batches in flight on separate HIP streams (a serving loop's option, NOT the contract's step: two batches
        # of B frames are resident at once, so this never becomes `value`).  The lifter's 17-token kernels, the low-resolution
        # branches and every launch's tail leave CUs idle that the other batch's convolutions fill.
        overlapped = None
        if not a.train and world == 1 and a.overlap > 1:
            try:
                lanes = [(model, kc_work, torch.cuda.Stream(dev))]
                for _ in range(a.overlap - 1):
                    with contextlib.redirect_stdout(io.StringIO()):
                        m2 = CA_PF(cfg, compute_dtype="bf16" if a.dtype == "bf16" else "fp32", plan_flags=pflags).eval()
                    m2.load_state_dict(sd_cpu)
                    lanes.append((m2.to(dev), kc0.clone(), torch.cuda.Stream(dev)))
                if a.lanes >= 0:
                    for m, _, _ in lanes[1:]:
                        m.engine_for(img).set_lanes(a.lanes)

                def lane_step(i):
                    m, kcw, st = lanes[i % len(lanes)]
                    with torch.cuda.stream(st):
                        kcw.copy_(kc0)
                        return m(img, k2d, kcw)

                with torch.no_grad():
                    outs = [lane_step(i) for i in range(max(a.warmup, len(lanes)))]
                    fence()
                    same = all(torch.equal(o, out) for o in outs[-len(lanes):])       # every engine reproduces the contract step's output
                    t1 = time.perf_counter()
                    for i in range(a.steps):
                        lane_step(i)
                    fence()
                    el2 = time.perf_counter() - t1
                overlapped = {"streams": len(lanes), "value": round(B * a.steps / el2, 2), "unit": "frames/s", "steps": a.steps,
                              "ms_per_step": round(el2 / a.steps * 1e3, 4), "frames_in_flight": B * len(lanes),
                              "outputs_bit_identical_to_contract_step": bool(same),
                              "note": "same K steps, issued round-robin on separate engines / HIP streams; not the headline"}
                del lanes, outs
            except Exception as e:                                 # an extra measurement must never cost the contract's line
                overlapped = {"error": f"{type(e).__name__}: {e}"[:300]}
                torch.cuda.synchronize(dev)
        return overlapped

    alt_outputs = {}

    def measure_alt_plan(flag, key, note):
        # ---- fp32 configurations: the same K steps on another arithmetic route for the 3x3 convs, so that all of them are on one line (never `value`):
        # CAPF_PLAN_NO_F32X3 = round 3's plan, the fp32 matrix pipe (Winograd / direct kernels); CAPF_PLAN_F32X3_EXACT = round 4's plan, the
        # exact three-bf16-piece tile (six piece products per fp32 product where the product plan issues three)
        if a.train or world != 1 or a.dtype == "bf16" or a.no_f32x3 or a.no_alt_plan or (pflags & flag):
            return None
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                m3 = CA_PF(cfg, compute_dtype="fp32", plan_flags=(pflags & ~PLAN_F32X3_EXACT) | flag).eval()
            m3.load_state_dict(sd_cpu)
            m3 = m3.to(dev)
            if a.lanes >= 0:
                m3.engine_for(img).set_lanes(a.lanes)
            kc3 = kc0.clone()
            with torch.no_grad():
                for _ in range(max(a.warmup, 2)):
                    kc3.copy_(kc0)
                    o3 = m3(img, k2d, kc3)
                torch.cuda.synchronize(dev)
                t3 = time.perf_counter()
                for _ in range(a.steps):
                    kc3.copy_(kc0)
                    o3 = m3(img, k2d, kc3)
                torch.cuda.synchronize(dev)
                el3 = time.perf_counter() - t3
            alt_outputs[key] = o3.clone()
            res = {"value": round(B * a.steps / el3, 2), "unit": "frames/s", "ms_per_step": round(el3 / a.steps * 1e3, 4), "steps": a.steps,
                   "max_abs_diff_to_contract_step": float((o3 - out).abs().max()), "note": note}
            del m3, o3
            return res
        except Exception as e:                                 # an extra measurement must never cost the contract's line
            torch.cuda.synchronize(dev)
            return {"error": f"{type(e).__name__}: {e}"[:300]}

    dist_info = None
    if world > 1:
        # evidence that the job really ran on `world` ranks of the named backend: a SUM all-reduce of ones on the device
        # (through RCCL when the backend is nccl) and every rank's own rate
        on_host = dist.get_backend() == "gloo"
        ones = torch.ones(1, dtype=torch.float32, device="cpu" if on_host else dev)
        dist.all_reduce(ones)
        mine = torch.tensor([B * a.steps / own_elapsed], dtype=torch.float64, device="cpu" if on_host else dev)
        rates = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(rates, mine)
        dist_info = {"backend": dist.get_backend(), "ranks_seen": int(ones.item()),
                     "devices_visible": torch.cuda.device_count(),
                     "per_rank_frames_per_s": [round(r.item(), 2) for r in rates]}
    elapsed = cdist.max_over_ranks(elapsed, dev)
    ms_per_step = elapsed / a.steps * 1e3
    fps = global_frames * a.steps / elapsed

    result = None
    if rank == 0:
        eng = model.engine_for(img)
        # ---- per-kernel event timing on the launch stream (separate, untimed passes over the FORWARD schedule; in
        # the training configuration the frozen-backbone forward is 90 % of the step and owns the dominant kernel)
        table = eng.op_table(B)
        obytes = eng.op_bytes(B)
        oexec = eng.op_executed_flops(B)
        acc = {}
        out_buf = torch.empty_like(out)
        # A launch's duration is the difference of the event markers in front of it and in front of the next one: it contains one
        # marker's own cost on the GPU's command processor, which is not the kernel's.  Calibrated here (markers back to back on the
        # same stream, nothing between them) and subtracted from every launch: with it the per-kernel averages agree with rocprofv3's
        # kernel durations (profiles/), without it they sit 5-9 us above them.
        cal_ev = [torch.cuda.Event(enable_timing=True) for _ in range(129)]
        for e in cal_ev:
            e.record(stream)
        torch.cuda.synchronize(dev)
        marker_ms = sorted(cal_ev[i].elapsed_time(cal_ev[i + 1]) for i in range(128))[64]
        with torch.no_grad():
            for _ in range(max(1, a.profile_steps)):
                kc_work.copy_(kc0)
                # the PRODUCT schedule, one event pair per launch (a grouped launch carries several convs)
                ms, leader = eng.forward_profile_launches(img, k2d, kc_work, out_buf, stream.cuda_stream)
                variants = eng.profile_variants()
                members = {}
                for i, l in enumerate(leader):
                    if l >= 0:
                        members.setdefault(l, []).append(i)
                for l, ops_ in members.items():
                    kern = table[l][1]
                    if len(ops_) > 1 and kern.startswith("igemm"):
                        # (the grouped bf16 launch has three device kernels: name the one this launch ran, as rocprofv3 will)
                        kern = (("igemm_bf16_group", "igemm_bf16_group_pp", "igemm_bf16_group_rh", "igemm_bf16_group_ws")[max(0, variants[l])] if kern.startswith("igemm_bf16")
                                else ("igemm_wino43_group" if kern.startswith("igemm_wino43") else
                                      "igemm_wino_group" if kern.startswith("igemm_wino") else
                                      "igemm_f32h2g_group" if kern.startswith("igemm_f32h2g") else
                                      kern if kern.startswith(("igemm_f32_pwchain", "igemm_f32x3", "igemm_f32h2")) else "igemm_f32_group"))
                        if table[l][1].startswith("igemm_bf16_pwchain"):
                            kern = table[l][1]
                    if not kern or table[l][0].startswith("copy."):
                        continue
                    e = acc.setdefault(kern, [0.0, 0.0, 0, 0.0, 0.0])
                    e[0] += max(ms[l] - marker_ms, 0.0); e[1] += sum(table[i][2] for i in ops_); e[2] += 1; e[3] += sum(obytes[i] for i in ops_)
                    e[4] += sum(oexec[i] for i in ops_)
                n_launches = len(members)
        nprof = max(1, a.profile_steps)
        total_ms = sum(e[0] for e in acc.values())
        dname, (dms, dflops, dn, dbytes, dexec) = max(acc.items(), key=lambda kv: kv[1][0])
        tflops = dflops / (dms * 1e-3) / 1e12 if dms > 0 else 0.0
        gbs = dbytes / (dms * 1e-3) / 1e9 if dms > 0 else 0.0
        gemm_ms = sum(e[0] for k, e in acc.items() if k.startswith("igemm"))
        gemm_fl = sum(e[1] for k, e in acc.items() if k.startswith("igemm"))
        gemm_by = sum(e[3] for k, e in acc.items() if k.startswith("igemm"))
        gemm_ex = sum(e[4] for k, e in acc.items() if k.startswith("igemm"))
        peak = PEAK_TFLOPS[a.dtype]
        # The split-fp32 tile (igemm_f32x3_ws.hip) computes fp32 results on the bf16 pipe with six piece products per fp32 product: its
        # roof in ALGORITHMIC fp32 FLOP/s is the bf16 peak / 6, so that frac = executed bf16 FLOP/s / bf16 peak = the pipe's busy fraction
        # ... the two-fp16-piece tile (igemm_f32h2_ws.hip) issues three: bf16 / fp16 peak / 3
        x3 = dname.startswith(("igemm_f32x3", "igemm_f32h2"))
        pieces = lambda k: 6.0 if k.startswith("igemm_f32x3") else 3.0
        dpeak = round(PEAK_TFLOPS["bf16"] / pieces(dname), 1) if x3 else peak
        pipe_peak = lambda k: PEAK_TFLOPS["bf16"] if k.startswith(("igemm_f32x3", "igemm_f32h2", "igemm_bf16")) else peak     # executed FLOPs are priced on the pipe they ran on
        gemm_busy = sum(e[4] / pipe_peak(k) for k, e in acc.items() if k.startswith("igemm")) / 1e12
        alg_peak = lambda k: PEAK_TFLOPS["bf16"] / pieces(k) if k.startswith(("igemm_f32x3", "igemm_f32h2")) else peak          # roof of a kernel in algorithmic FLOP/s
        gemm_alg = sum(e[1] / alg_peak(k) for k, e in acc.items() if k.startswith("igemm")) / 1e12
        # HBM bytes per launch of that kernel: offline PMC passes of this same command (FETCH_SIZE and WRITE_SIZE in
        # separate rocprofv3 --pmc runs, tools/summarize_profiles.py), null if not collected for this configuration
        traffic, tsrc, tstale = None, None, None
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        try:
            from summarize_profiles import csrc_sha
            sha_now = csrc_sha()
        except Exception:
            sha_now = None
        for rnd in ("r06", "r05", "r04", "r03", "r02"):
            tfile = os.path.join(ROOT, "profiles", f"{rnd}_hbm_traffic.json")
            if traffic is None and tstale is None and tag is not None and os.path.exists(tfile):
                data = json.load(open(tfile)).get(f"cfg{tag}", {})
                ent = data.get(dname) if isinstance(data.get(dname), dict) else {}
                # a "launch" of this line is what `launches_per_step` counts -- one grouped launch of a dependency LEVEL, which at batch
                # 512 is TWO kernel launches (64- and 32-channel tiles): prefer the file's bytes per FORWARD over this run's levels per
                # forward; an older file only has the per-kernel-launch average
                got = (ent["hbm_bytes_per_forward"] / max(1, dn // nprof)) if ent.get("hbm_bytes_per_forward") else ent.get("hbm_bytes_per_launch")
                if got is None:
                    continue
                sha_then = data.get("_csrc_sha")
                if sha_then is not None and sha_now is not None and sha_then != sha_now:
                    tstale = (f"{os.path.relpath(tfile, ROOT)} was collected on csrc {sha_then}, this run times csrc {sha_now}: "
                              f"not reported (re-run tools/profile_round.sh)")
                    break
                traffic = got
                tsrc = (os.path.relpath(tfile, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes" +
                        (f"; csrc {sha_then}" if sha_then else "; no source hash recorded: an earlier round's file") + ")")
        mfma = {"bound": "mfma", "achieved": round(tflops, 2), "peak": dpeak, "unit": "TFLOP/s", "frac": round(tflops / dpeak, 4)}
        if x3:
            mfma["peak_note"] = ("fp32 results on the bf16 matrix pipe: each fp32 operand = three bf16 pieces (exact), six piece products per fp32 product; "
                                 "peak = 2500 TFLOP/s dense bf16 / 6" if dname.startswith("igemm_f32x3") else
                                 "fp32 results on the fp16 matrix pipe: each fp32 operand = two block-scaled fp16 pieces (to 2^-23), three piece products per "
                                 "fp32 product; peak = 2500 TFLOP/s dense fp16 / 3, so frac = executed fp16 FLOP/s / the fp16 peak")
        hbm = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
        first, second = (mfma, hbm) if mfma["frac"] >= hbm["frac"] else (hbm, mfma)
        roofline = dict(first)
        roofline.update({
            "kernel": dname, "traffic": traffic, "traffic_source": tsrc if traffic else tstale, "other_roof": second,
            # `achieved` / `frac` above credit the kernel with the ALGORITHMIC (direct-convolution-equivalent) FLOPs, as SURVEY.md
            # 8(d) prescribes; a Winograd kernel EXECUTES 1/2 (F(4,3)) or 2/3 (F(2,3)) of those multiplies, so the matrix pipe's
            # own utilisation is the second pair: executed FLOPs / duration / peak (an upper bound on SQ_VALU_MFMA_BUSY, which the
            # PMC pass in profiles/ measures directly)
            "flops_convention": "algorithmic = 2*M*N*K of the direct convolution (SURVEY 8d); executed = MFMA MACs actually issued",
            "executed_flops_per_launch": round(dexec / dn, 1),
            "executed_tflops": round(dexec / (dms * 1e-3) / 1e12, 2) if dms > 0 else 0.0,
            "mfma_busy_frac": round(dexec / (dms * 1e-3) / 1e12 / pipe_peak(dname), 4) if dms > 0 else 0.0,
            "algorithmic_flops_per_launch": round(dflops / dn, 1), "algorithmic_bytes_per_launch": round(dbytes / dn, 1),
            "launches_per_step": dn // nprof, "avg_launch_us": round(dms / dn * 1e3, 2),
            "share_of_forward": round(dms / total_ms, 4),
            "measured_hbm_gbs": round(traffic / (dms / dn * 1e-3) / 1e9, 1) if traffic else None,
            "all_mfma_kernels": {"tflops": round(gemm_fl / (gemm_ms * 1e-3) / 1e12, 2),
                                 "mfma_frac": round(gemm_alg / (gemm_ms * 1e-3), 4),      # each kernel's algorithmic FLOPs over its own roof
                                 "mfma_busy_frac": round(gemm_busy / (gemm_ms * 1e-3), 4),
                                 "algorithmic_gbs": round(gemm_by / (gemm_ms * 1e-3) / 1e9, 1),
                                 "share_of_forward": round(gemm_ms / total_ms, 4)},
            "forward_ms_by_events": round(total_ms / nprof, 3), "event_marker_us": round(marker_ms * 1e3, 2)})
        if a.kernel_table:
            for k, e in sorted(acc.items(), key=lambda kv: -kv[1][0]):
                tf = e[1] / (e[0] * 1e-3) / 1e12 if e[0] > 0 else 0
                gb = e[3] / (e[0] * 1e-3) / 1e9 if e[0] > 0 else 0
                ex = e[4] / (e[0] * 1e-3) / 1e12 if e[0] > 0 else 0
                print(f"  {k:34s} {e[2] // nprof:4d} launches/step {e[0] / nprof:9.3f} ms/step {tf:8.2f} TFLOP/s(alg) {ex:8.2f} TFLOP/s(exec) "
                      f"{gb:8.1f} GB/s(alg)", file=sys.stderr)
        sustained = measure_sustained()
        overlapped = measure_overlapped()                     # (extra measurements run after the per-launch timing passes: they leave the chip warm)
        exact_split_plan = measure_alt_plan(PLAN_F32X3_EXACT, "exact_split_plan", "plan_flags |= CAPF_PLAN_F32X3_EXACT: the 3x3 convs on round 4's tile -- every operand "
                                            "split EXACTLY into three bf16 pieces, six piece products per fp32 product; not the headline")
        fp32_pipe_plan = measure_alt_plan(PLAN_NO_F32X3 | PLAN_NO_F32H2_GEMM, "fp32_pipe_plan", "plan_flags |= CAPF_PLAN_NO_F32X3 | CAPF_PLAN_NO_F32H2_GEMM: every conv / GEMM on "
                                          "the fp32 matrix pipe (3x3: Winograd from batch 24, direct below) -- round 3's plan; not the headline")
        launches, flops = eng.stats(B)
        par = f"dp{world} (independent frames, " + ("one flat-gradient all-reduce per step)" if a.train else "no collective)")
        result = {
            "metric": "frames/sec", "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": workload_string(a, tag), "baseline_config": tag, "frames_per_step": global_frames, "frames_per_gpu": B,
                       "parallelism": par, "launches_per_forward": n_launches, "forward_gflop_per_frame": round(flops / B / 1e9, 3),
                       "plan_flags": pflags},
            "end_to_end_forward_tflops": round(fps * flops / B / 1e12, 2),
            "roofline": roofline,
        }
        if dist_info is not None:
            result["distributed"] = dist_info
        if sustained is not None:
            result["sustained"] = sustained
        if overlapped is not None:
            result["overlapped_steps"] = overlapped
        if exact_split_plan is not None:
            result["exact_split_plan"] = exact_split_plan
        if fp32_pipe_plan is not None:
            result["fp32_pipe_plan"] = fp32_pipe_plan
        if train_phases is not None:
            result["train_phases"] = train_phases
        if world == 1 and not a.no_cpu_baseline:
            if not a.train:
                result["vs_fp32_oracle"] = vs_fp32_oracle(a.backbone, sd_cpu, img, k2d, kc0, out, alt_outputs)
                if a.dtype == "bf16":
                    result["vs_fp32_oracle"]["reference_in_bf16"] = reference_in_bf16(a.backbone)
            result["cpu_baseline"] = cpu_baseline(a.backbone, H, W, sd_cpu)
            result["gpu_over_cpu"] = round(fps / result["cpu_baseline"]["value"], 1)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
