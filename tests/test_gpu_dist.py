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


def _worker(rank, world, port, q):
    for p in (os.path.join(ROOT, "contextaware-poseformer_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from capf import dist as cd, synth
    from conftest import make_model
    from mvn.models.loss import MPJPE
    cd.init_from_env("gloo")
    torch.cuda.set_device(0)
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
