"""GPU box (ONE GPU): two ranks share cuda:0 and all-reduce (gloo, CPU-staged — RCCL refuses two ranks on one device)
the REAL flat gradient buffer capf_backward writes; the average of the two half-batch gradients must equal the
single-process gradient of the concatenated batch (MPJPE is a mean over equal shards), and
bench.py's own --train N=2 path must run end to end.  No scaling number comes out of this: it is a correctness
check of the training configuration's only exchange step (train.py:195, :361-362)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q, backend="gloo", one_device=True):
    for p in (os.path.join(ROOT, "contextaware-poseformer_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0" if one_device else str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from capf import dist as cd, synth
    from conftest import make_model
    from mvn.models.loss import MPJPE
    torch.cuda.set_device(0 if one_device else rank)
    cd.init_from_env(backend)
    model, _ = make_model("hrnet_32", device="cuda", wseed=11)
    model.train(); model.backbone.eval(); model.drop_path_rate = 0.0
    cd.broadcast_state_(model.volume_net)
    B = 8
    img, k2d, kc, gt = synth.synth_inputs(B, 256, 192, seed=12, with_gt=True)

    def flat_grad(lo, hi):
        model.zero_grad(set_to_none=True)
        pred = model(img[lo:hi].cuda(), k2d[lo:hi].cuda(), kc[lo:hi].clone().cuda())
        MPJPE()(pred, gt[lo:hi].cuda()).backward()
        return model.last_flat_grad.clone()

    lo, hi = cd.shard_bounds(B, rank, world)
    mine = flat_grad(lo, hi)
    cd.allreduce_mean_(mine)
    whole = flat_grad(0, B)
    torch.cuda.synchronize()
    scale = whole.abs().max().item()
    err = (mine - whole).abs().max().item()
    cd.barrier()
    q.put((rank, err, scale, mine.numel()))
    torch.distributed.destroy_process_group()


def test_two_ranks_allreduce_the_real_flat_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, err, scale, n in res:
        print(f"rank {rank}: max |mean of shard grads - full-batch grad| = {err:.3e} (grad max {scale:.3e}, {n} elements)")
        assert n == 14094147 and err <= 2e-5 * scale + 1e-9


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="a REAL multi-rank RCCL all-reduce needs two visible devices (RCCL refuses two ranks on one)")
def test_two_ranks_allreduce_the_real_flat_gradient_over_rccl():
    """wherever two devices are visible (the driver's 8-GPU node), the first multi-rank RCCL execution of this repository is a
    TEST, not the scaling bench: one rank per device, backend nccl (= RCCL, xGMI between the devices), the same check as the gloo test above --
    the mean of the two half-batch flat gradients that capf_backward wrote equals the full-batch gradient."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, "nccl", False)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, err, scale, n in res:
        print(f"RCCL rank {rank}: max |mean of shard grads - full-batch grad| = {err:.3e} (grad max {scale:.3e}, {n} elements)")
        assert n == 14094147 and err <= 2e-5 * scale + 1e-9


def test_bench_train_two_ranks_on_one_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CAPF_BENCH_SINGLE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--train", "--batch", "16", "--steps", "3",
                        "--warmup", "1", "--backend", "gloo", "--profile-steps", "1"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["config"]["frames_per_step"] == 32 and j["value"] > 0
    assert "all-reduce" in j["config"]["parallelism"] and j["config"]["baseline_config"] is None


def _rccl_worker(port, q):
    for p in (os.path.join(ROOT, "contextaware-poseformer_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from capf import dist as cd, synth
    from capf.optim import FusedAdamW, flatten_
    from conftest import make_model
    from mvn.models.loss import MPJPE
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method="env://", rank=0, world_size=1)      # "nccl" IS RCCL on ROCm
    model, _ = make_model("hrnet_32", device="cuda", wseed=11)
    model.train(); model.backbone.eval(); model.drop_path_rate = 0.0
    cd.broadcast_state_(model.volume_net)                      # world 1: no-op by construction
    flat_p = flatten_(model.volume_net)
    opt = FusedAdamW(flat_p, lr=1e-3, weight_decay=0.1)
    model.flat_grad_only = True
    img, k2d, kc, gt = synth.synth_inputs(4, 256, 192, seed=12, with_gt=True)
    pred = model(img.cuda(), k2d.cuda(), kc.clone().cuda())
    MPJPE()(pred, gt.cuda()).backward()
    g = model.last_flat_grad
    before = g.clone()
    dist.all_reduce(g, op=dist.ReduceOp.SUM)                   # the REAL 56.4 MB buffer through an RCCL collective on the device
    torch.cuda.synchronize()
    same = torch.equal(g, before)
    g2, scale = cd.allreduce_sum_(g)                           # the bench step's own path (skips the collective at world 1)
    p0 = flat_p.clone()
    opt.step(g2, grad_scale=scale)
    torch.cuda.synchronize()
    moved = (flat_p - p0).abs().max().item()
    q.put((dist.get_backend(), g.numel(), same, scale, moved))
    dist.destroy_process_group()


def test_rccl_runs_an_allreduce_of_the_real_flat_gradient_single_rank():
    """The only RCCL evidence a ONE-GPU box can give: the library loads, a communicator initialises (world size 1) and an
    all-reduce of the real 14.09 M-element flat gradient that capf_backward wrote runs on the device and leaves it unchanged.
    Multi-rank RCCL over xGMI is the driver's 8-GPU run (bench.py records backend / ranks_seen / per-rank rates for it)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    backend, n, same, scale, moved = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert backend == "nccl" and n == 14094147 and same and scale == 1.0 and moved > 0
