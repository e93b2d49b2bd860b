#!/usr/bin/env python
"""Headline benchmark: frames/sec of the Context-Aware PoseFormer hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" is ONE pass of the hot path (CA_PF.forward through the C ABI: HRNet backbone -> joint-context
sampling -> lifting transformer) over one batch of synthetic input that is already resident in HBM.
Workload at every N: BASELINE.json configs[1] — batch 64 per GPU, HRNet-32, 256x256, fp32 (weak
scaling: frames are independent, ranks share nothing, no data-path collective — SURVEY.md §8e).
Rank 0 prints ONE JSON line with the whole-job frames/s plus
  roofline     — for the dominant kernel: algorithmic FLOPs of its launches / their summed duration,
                 measured with HIP event pairs on the launch stream (capf_forward_profile), against
                 the dense fp32-MFMA peak of MI355X_MICROARCH.md (157.3 TFLOP/s);
  cpu_baseline — the CPU oracle (a port of the reference forward to functional PyTorch-CPU) timed
                 on this box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "contextaware-poseformer_amd"))

PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}          # MI355X_MICROARCH.md: dense fp32-input MFMA (= vector) peak
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (default 64; 512 with --train)")
    ap.add_argument("--train", action="store_true", help="BASELINE.json configs[3]: training step (frozen backbone forward, "
                    "lifter forward+backward, MPJPE, gradient all-reduce, fused AdamW) instead of inference")
    ap.add_argument("--backbone", default="hrnet_32", choices=["hrnet_32", "hrnet_48", "cpn"])
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--kernel-table", action="store_true", help="print the per-kernel table to stderr")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="bf16: backbone convs on bf16 MFMA (configs[2]/[4])")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo for smoke tests)")
    ap.add_argument("--lanes", type=int, default=-1,
                    help="independent backbone branches: 2 grouped launches (default), 1 side streams, 0 program order")
    return ap.parse_args()


def cpu_baseline(backbone, H, W, sd_cpu, budget_s=15.0):
    """Time the CPU oracle on a bounded sample of the workload (batch-8 forwards, <= ~budget_s of CPU
    work).  Thread count: oneDNN convs stop scaling (and can collapse) far below the core count of a
    256-core host, so 16/32/64 threads are tried in turn while they keep paying; `cores` in the result
    is the thread count actually used for the reported number."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import capf_oracle as oracle
    from capf import synth
    host = os.cpu_count() or 1
    B = 8
    img, k2d, kc = synth.synth_inputs(B, H, W, seed=101)

    def run(threads, reps):
        torch.set_num_threads(threads)
        ts = []
        with torch.no_grad():
            for _ in range(reps):
                t0 = time.perf_counter()
                oracle.ca_pf_forward(sd_cpu, img, k2d, kc.clone(), backbone=backbone)
                ts.append(time.perf_counter() - t0)
        return ts

    t_start = time.perf_counter()
    best_threads, best = None, None
    for threads in [t for t in (16, 32, 64) if t <= host] or [host]:
        ts = run(threads, 2)                       # first call doubles as warm-up
        t = min(ts)
        if best is None or t < best:
            best_threads, best = threads, t
        if t > 1.25 * best or time.perf_counter() - t_start > budget_s / 2:
            break
    times = run(best_threads, 1)
    while time.perf_counter() - t_start < budget_s and len(times) < 30:
        times += run(best_threads, 1)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(B / med, 2), "unit": "frames/s", "cores": best_threads, "host_cores": host, "kind": "port",
            "sample": f"{len(times)} x batch-{B} {backbone} {H}x{W} fp32 forwards of oracle/capf_oracle.py "
                      f"(functional PyTorch-CPU/oneDNN, {best_threads} threads), median"}


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from capf import synth
    from mvn.models.conpose import CA_PF
    from mvn.utils.cfg import backbone_preset, config

    from capf import dist as cdist
    rank, world, local = cdist.init_from_env(a.backend)    # "nccl" is RCCL on ROCm; no-op when WORLD_SIZE == 1
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if os.environ.get("CAPF_BENCH_SINGLE_DEVICE"):         # smoke-testing the N>1 flow on a 1-GPU box (gloo only)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    cfg = backbone_preset(copy.deepcopy(config), a.backbone)
    cfg.model.backbone.fix_weights = True
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        model = CA_PF(cfg, compute_dtype="bf16" if a.dtype == "bf16" else "fp32").eval()
    sd_cpu = synth.load_synthetic(model, seed=1, bn_mode="random")
    model = model.to(dev)

    B, H, W = a.batch or (512 if a.train else 64), a.height, a.width
    img, k2d, kc, gt = synth.synth_inputs(B, H, W, seed=1000 + rank, crop_range=(192, 256), with_gt=True)
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
        crit = MPJPE()

        def step():
            kc_work.copy_(kc0)
            pred = model(img, k2d, kc_work)                                  # DropPath active (dpr 0..0.2)
            loss = crit(pred, gt)
            model.zero_grad(set_to_none=True)
            loss.backward()
            flat_g = model.last_flat_grad
            cdist.allreduce_mean_(flat_g)                                    # ONE RCCL all-reduce of 56.4 MB (C3)
            opt.step(flat_g)
            model.lifter_params_changed()
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
        fence()
        elapsed = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    elapsed = cdist.max_over_ranks(elapsed, dev)
    ms_per_step = elapsed / a.steps * 1e3
    fps = B * world * a.steps / elapsed

    result = None
    if rank == 0:
        eng = model.engine_for(img)
        # ---- per-kernel event timing on the launch stream (separate, untimed passes)
        table = eng.op_table(B)
        acc = {}
        out_buf = torch.empty_like(out)
        with torch.no_grad():
            for _ in range(max(1, a.profile_steps)):
                kc_work.copy_(kc0)
                # the PRODUCT schedule, one event pair per launch (a grouped launch carries several convs)
                ms, leader = eng.forward_profile_launches(img, k2d, kc_work, out_buf, stream.cuda_stream)
                members = {}
                for i, l in enumerate(leader):
                    if l >= 0:
                        members.setdefault(l, []).append(i)
                for l, ops_ in members.items():
                    kern = table[l][1] if len(ops_) == 1 else ("igemm_bf16_group" if table[l][1].startswith("igemm_bf16") else "igemm_f32_group")
                    if not kern or table[l][0].startswith("copy."):
                        continue
                    e = acc.setdefault(kern, [0.0, 0.0, 0])
                    e[0] += ms[l]; e[1] += sum(table[i][2] for i in ops_); e[2] += 1
                n_launches = len(members)
        total_ms = sum(e[0] for e in acc.values())
        dom = max(acc.items(), key=lambda kv: kv[1][0])
        dname, (dms, dflops, dn) = dom
        achieved = dflops / (dms * 1e-3) / 1e12 if dms > 0 else 0.0
        gemm_ms = sum(e[0] for k, e in acc.items() if k.startswith("igemm"))
        gemm_fl = sum(e[1] for k, e in acc.items() if k.startswith("igemm"))
        peak = PEAK_TFLOPS[a.dtype]
        # HBM bytes per launch of that kernel: offline PMC passes of this same command (FETCH_SIZE and
        # WRITE_SIZE in separate rocprofv3 --pmc runs, tools/summarize_profiles.py), null if not collected
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        if os.path.exists(tfile) and (a.backbone, B, H, W, a.train) == ("hrnet_32", 64, 256, 256, False):
            traffic = json.load(open(tfile)).get(dname, {}).get("hbm_bytes_per_launch")
        roofline = {"bound": "mfma", "kernel": dname, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": traffic,
                    "traffic_source": "profiles/r01_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)" if traffic else None,
                    "algorithmic_flops_per_launch": round(dflops / dn, 1),
                    "launches_per_step": dn // max(1, a.profile_steps),
                    "avg_launch_us": round(dms / dn * 1e3, 2),
                    "share_of_step": round(dms / total_ms, 4),
                    "all_mfma_kernels": {"achieved": round(gemm_fl / (gemm_ms * 1e-3) / 1e12, 2),
                                         "frac": round(gemm_fl / (gemm_ms * 1e-3) / 1e12 / peak, 4),
                                         "share_of_step": round(gemm_ms / total_ms, 4)}}
        if a.kernel_table:
            for k, e in sorted(acc.items(), key=lambda kv: -kv[1][0]):
                n = e[2] // max(1, a.profile_steps)
                tf = e[1] / (e[0] * 1e-3) / 1e12 if e[0] > 0 else 0
                print(f"  {k:34s} {n:4d} launches/step {e[0] / max(1, a.profile_steps):9.3f} ms/step {tf:7.2f} TFLOP/s",
                      file=sys.stderr)
        launches, flops = eng.stats(B)
        result = {
            "metric": "frames/sec", "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": (f"configs[3]: TRAINING step, batch {B}/GPU {a.backbone} {H}x{W} (frozen backbone forward, lifter "
                                    f"fwd+bwd, MPJPE, flat-gradient all-reduce, fused AdamW, DropPath on), fp32" if a.train else
                                    f"configs[1]: batch {B}/GPU {a.backbone} {H}x{W} image + 17 kpts -> 17x3, "
                                    f"PoseFormer lifter embed 128 levels 4, fp32 inference"),
                       "frames_per_step": B * world, "parallelism": f"dp{world} (independent frames, no collective)",
                       "launches_per_step": n_launches, "gflop_per_frame": round(flops / B / 1e9, 3)},
            "end_to_end_tflops": round(fps * flops / B / 1e12, 2),
            "roofline": roofline,
        }
        if world == 1 and not a.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(a.backbone, H, W, sd_cpu)
            result["gpu_over_cpu"] = round(fps / result["cpu_baseline"]["value"], 1)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
