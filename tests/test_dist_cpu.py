"""CPU, world_size 2 over gloo: the multi-process pieces of the data-parallel path —
frame sharding with no data-path collective, the padded result gather (train.py:216-226), gradient
averaging == single-process gradient on the concatenated batch (MPJPE is a mean over equal shards),
parameter broadcast, max-over-ranks timing."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "contextaware-poseformer_amd"))
    sys.path.insert(0, os.path.join(root, "oracle"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from capf import dist as cd
    import capf_oracle as oracle
    r, w, _ = cd.init_from_env("gloo")
    assert (r, w) == (rank, world)

    # ---- sharding + gather of per-frame results (odd total: remainder goes to the last rank)
    total = 7
    lo, hi = cd.shard_bounds(total, rank, world)
    full = torch.arange(total * 17 * 3, dtype=torch.float32).view(total, 1, 17, 3)
    got = cd.gather_predictions(full[lo:hi].clone(), total)
    ok_gather = torch.equal(got, full)

    # ---- gradient averaging: tiny lifter-like model, MPJPE loss, equal shards
    torch.manual_seed(0)
    lin = torch.nn.Linear(6, 3)
    cd.broadcast_state_(lin)
    x = torch.randn(8, 1, 17, 6, generator=torch.Generator().manual_seed(1))
    gt = torch.randn(8, 1, 17, 3, generator=torch.Generator().manual_seed(2))
    a, b = cd.shard_bounds(8, rank, world)
    loss = oracle.mpjpe(lin(x[a:b]), gt[a:b])
    loss.backward()
    flat = torch.cat([p.grad.reshape(-1) for p in lin.parameters()])
    cd.allreduce_mean_(flat)
    ref = torch.nn.Linear(6, 3)
    ref.load_state_dict(lin.state_dict())
    oracle.mpjpe(ref(x), gt).backward()
    flat_ref = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    ok_grad = torch.allclose(flat, flat_ref, atol=1e-6)

    # ---- the same with the REAL lifter (oracle restatement of pose_dformer.py) and the flat gradient buffer laid out
    # exactly as capf_backward writes it (capf_grad_info order), on small context maps: 2 ranks x 2 frames
    import copy
    from capf import synth
    from capf.lib import Engine
    from mvn.models import _native
    from mvn.utils.cfg import backbone_preset, config
    cfg = backbone_preset(copy.deepcopy(config), "hrnet_32")
    plan = Engine(_native.make_capf_config(cfg, 256, 192), device=None)
    layout, total = plan.grad_layout()
    shapes = {n: s for n, s, k in plan.schema() if n in layout}
    plan.close()
    P = {n: synth.synth_tensor(n, shp, 5, shapes) for n, shp in shapes.items()}
    P = {n: (torch.from_numpy(v) if not torch.is_tensor(v) else v).clone().requires_grad_(True) for n, v in P.items()}
    g = torch.Generator().manual_seed(3)
    Bt = 4
    feats = [torch.randn(Bt, c, 16 >> i, 12 >> i, generator=g) for i, c in enumerate((32, 64, 128, 256))]
    k2d = torch.rand(Bt, 17, 2, generator=g) * 2 - 1
    ref_pts = torch.rand(Bt, 17, 2, generator=g) * 2 - 1
    gt3 = torch.randn(Bt, 1, 17, 3, generator=g) * 0.3

    def flat_grad(lo_, hi_):
        for v in P.values():
            v.grad = None
        pred = oracle.lifter_forward(P, k2d[lo_:hi_], ref_pts[lo_:hi_], [f[lo_:hi_] for f in feats])
        oracle.mpjpe(pred, gt3[lo_:hi_]).backward()
        buf = torch.zeros(total)
        for n, (off, cnt) in layout.items():
            buf[off:off + cnt] = P[n].grad.reshape(-1)
        return buf

    a2, b2 = cd.shard_bounds(Bt, rank, world)
    mine = flat_grad(a2, b2)
    cd.allreduce_mean_(mine)                      # ONE collective over the 14.09 M-element buffer
    whole = flat_grad(0, Bt)
    ok_grad = ok_grad and total == 14094147 and torch.allclose(mine, whole, atol=2e-6, rtol=1e-4)

    t = cd.max_over_ranks(1.0 + rank, torch.device("cpu"))
    cd.barrier()
    q.put((rank, ok_gather, ok_grad, t))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_pipeline():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    assert all(r[1] for r in res), "padded all_gather of predictions differs from the full tensor"
    assert all(r[2] for r in res), "averaged 2-rank gradients differ from the single-process gradient"
    assert all(abs(r[3] - 2.0) < 1e-9 for r in res)


def test_shard_bounds_cover_everything():
    import sys
    from capf import dist as cd
    for n in (1, 7, 64, 513):
        for w in (1, 2, 4, 8):
            spans = [cd.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


def test_bench_self_spawns_two_ranks_dry_run():
    """`python bench.py --gpus 2` WITHOUT a launcher re-executes itself under torch.distributed.run (two ranks on
    127.0.0.1); --dry-run replaces the GPU step by a sleep, everything else (rendezvous, barriers, max-over-ranks, one
    JSON line from rank 0) is bench.py's own N>1 code path."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--dry-run",
                        "--backend", "gloo", "--config", "3"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["dry_run"] and j["config"]["frames_per_step"] == 1024 and j["scaling"] == "weak"
    assert j["ms_per_step"] >= 4.0            # max over ranks: rank 1 sleeps 4 ms per step
    assert j["config"]["workload"].startswith("configs[3]: TRAINING")
    # evidence a driver-side SCALE record can be checked against: the backend that ran, how many ranks a collective on it
    # saw, and every rank's own rate (the GPU path emits the same keys, measured through the data-parallel backend)
    d = j["distributed"]
    assert d["backend"] == "gloo" and d["ranks_seen"] == 2 and len(d["per_rank_frames_per_s"]) == 2
    assert d["per_rank_frames_per_s"][0] > d["per_rank_frames_per_s"][1] > 0     # rank 1 sleeps twice as long


def test_bench_eight_rank_dry_run_of_the_cpn_configuration():
    """The flow the driver's 8-GPU scaling run takes for configs[4] (`--gpus 8 --config 4`), as a dry run on CPU: eight ranks
    rendezvous on 127.0.0.1, a collective on the backend sees all eight, every rank reports its own rate and draws its own input seed,
    rank 0 prints one line whose frames_per_step is the whole job's (weak scaling: 128 frames per rank)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--dry-run",
                        "--backend", "gloo", "--config", "4"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    d = j["distributed"]
    assert j["n_gpus"] == 8 and j["config"]["frames_per_step"] == 8 * 128 and j["scaling"] == "weak" and j["dtype"] == "bf16"
    assert d["ranks_seen"] == 8 and len(d["per_rank_frames_per_s"]) == 8
    assert len(set(d["input_seeds"])) == 8
    assert j["config"]["workload"].startswith("configs[4]")


def test_bench_strong_scaling_dry_run_splits_a_fixed_global_batch():
    """BASELINE.md section 4 item 4 asks for weak AND global-batch-fixed scaling: `--scaling strong` keeps --batch as the GLOBAL batch and
    shards it over the ranks (capf.dist.shard_bounds), says so in the line, and counts the whole job's frames once."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1", "--dry-run",
                        "--backend", "gloo", "--config", "3", "--scaling", "strong"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["scaling"] == "strong" and j["n_gpus"] == 4
    assert j["config"]["frames_per_step"] == 512 and j["config"]["frames_per_gpu"] == 128 and j["config"]["shard_of_rank0"] == [0, 128]
    assert "GLOBAL batch 512 split over 4 GPUs" in j["config"]["workload"]
    assert abs(j["value"] - 512 * 3 / (j["ms_per_step"] * 3e-3)) < 1.0          # the job's frames over the slowest rank's time
    # a global batch that does not divide is refused, not rounded
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0", "--dry-run",
                        "--backend", "gloo", "--batch", "64", "--scaling", "strong"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "does not divide" in (r.stderr + r.stdout)


def test_bench_labels_follow_the_arguments():
    import importlib.util, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    a = bench.parse([])
    assert bench.config_tag(a) == 1 and "configs[1]" in bench.workload_string(a, 1) and a.dtype == "f32"
    a = bench.parse(["--backbone", "hrnet_48", "--batch", "256", "--dtype", "bf16"])
    assert bench.config_tag(a) == 2 and "bf16" in bench.workload_string(a, 2)
    a = bench.parse(["--train"])
    assert bench.config_tag(a) == 3 and a.batch == 512
    a = bench.parse(["--config", "4"])
    assert (a.backbone, a.height, a.width, a.dtype, a.batch) == ("cpn", 384, 288, "bf16", 128)
    a = bench.parse(["--batch", "32"])
    assert bench.config_tag(a) is None and bench.workload_string(a, None).startswith("custom")
