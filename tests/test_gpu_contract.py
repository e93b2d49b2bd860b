"""GPU box: the CALLER's contract of SURVEY.md §8(b), exercised the way ContextPose/train.py uses the model —
    torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)            train.py:317-318
    model.load_state_dict(<'module.'-stripped checkpoint>, strict)  train.py:309-312
    DistributedDataParallel(model, device_ids=[device])             train.py:361-362
    model.module.backbone.eval(); model.module.volume_net.train()   train.py:146-148
Two ranks share cuda:0 over gloo (RCCL refuses two ranks on one device).  The leaf modules of the host CA_PF only HOLD
parameters (their torch forward never runs), so what is tested here is that torch's machinery around them — module
conversion, state_dict round trips, DDP's reducer hooks firing through the native autograd node — still does its job."""
import os
import socket
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


def test_syncbn_conversion_keeps_the_state_dict_and_the_engine_binds():
    """convert_sync_batchnorm swaps the 292 BatchNorm2d leaves for SyncBatchNorm: same 1943 state_dict names / shapes /
    values, the engine binds the converted tree and computes the same joints bit for bit."""
    from capf import synth
    from conftest import make_model
    model, _ = make_model("hrnet_32", wseed=3)
    names = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    conv = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    assert conv is model                                                     # the root is not a BatchNorm: converted in place
    n_sync = sum(isinstance(m, torch.nn.SyncBatchNorm) for m in conv.modules())
    assert n_sync == 292 and not any(type(m) is torch.nn.BatchNorm2d for m in conv.modules())
    assert [(k, tuple(v.shape)) for k, v in conv.state_dict().items()] == names and len(names) == 1943
    plain, _ = make_model("hrnet_32", device="cuda", wseed=3)
    conv = conv.cuda().eval()
    img, k2d, kc = synth.synth_inputs(2, 256, 192, seed=4)
    with torch.no_grad():
        a = plain(img.cuda(), k2d.cuda(), kc.clone().cuda())
        b = conv(img.cuda(), k2d.cuda(), kc.clone().cuda())
    assert torch.equal(a, b)


def test_module_prefixed_checkpoint_loads_strict_after_the_callers_strip():
    """train.py:309-312: a checkpoint saved from the DDP-wrapped model carries 'module.' prefixes; the caller strips them
    and loads strict=True.  Values must reach the engine (the load hook marks it dirty)."""
    from capf import synth
    from conftest import make_model
    src, _ = make_model("hrnet_32", device="cuda", wseed=5)
    dst, _ = make_model("hrnet_32", device="cuda", wseed=6)
    img, k2d, kc = synth.synth_inputs(2, 256, 192, seed=7)
    with torch.no_grad():
        before = dst(img.cuda(), k2d.cuda(), kc.clone().cuda())            # engine bound to the OLD values
        want = src(img.cuda(), k2d.cuda(), kc.clone().cuda())
    checkpoint = {"module." + k: v.cpu() for k, v in src.state_dict().items()}   # what torch.save(ddp_model.state_dict()) holds
    for k in list(checkpoint.keys()):
        checkpoint[k.replace("module.", "")] = checkpoint.pop(k)
    ret = dst.load_state_dict(checkpoint, strict=True)
    assert not ret.missing_keys and not ret.unexpected_keys
    with torch.no_grad():
        got = dst(img.cuda(), k2d.cuda(), kc.clone().cuda())
    assert not torch.equal(before, want) and torch.equal(got, want)


def _ddp_worker(rank, world, port, q, sync_bn):
    for p in (os.path.join(ROOT, "contextaware-poseformer_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    from capf import dist as cd, synth
    from conftest import make_model
    from mvn.models.loss import MPJPE
    cd.init_from_env("gloo")
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    # rank 1 starts from DIFFERENT lifter weights: DDP's constructor broadcast (train.py:362, C1) must overwrite them
    model, _ = make_model("hrnet_32", wseed=11 + rank)
    if rank == 1:
        other, _ = make_model("hrnet_32", wseed=11)
        model.backbone.load_state_dict(other.backbone.state_dict())          # (frozen backbone: the same pretrained weights everywhere)
    if sync_bn:
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)        # train.py:317-318
    model = model.to(device)
    model.drop_path_rate = 0.0
    path = "native"
    try:
        ddp = DistributedDataParallel(model, device_ids=[device], output_device=0)        # train.py:361-362
    except RuntimeError as e:
        # this torch build's gloo has no device-tensor collectives: same wrapper, parameters synchronised by hand and the
        # reducer's buckets staged through the host by a comm hook -- the reducer / autograd-hook machinery is unchanged
        path = "gloo, host-staged buckets (" + str(e).splitlines()[0][:80] + ")"
        for t in model.volume_net.state_dict().values():
            h = t.detach().cpu()
            dist.broadcast(h, 0)
            t.copy_(h)
        ddp = DistributedDataParallel(model, device_ids=[device], output_device=0, init_sync=False, broadcast_buffers=False)

        def staged(state, bucket):
            buf = bucket.buffer()
            host = buf.cpu()
            dist.all_reduce(host)
            fut = torch.futures.Future()
            fut.set_result(host.to(buf.device).div_(world))
            return fut
        ddp.register_comm_hook(None, staged)
    ddp.train()
    ddp.module.backbone.eval()                                               # train.py:146-148
    ddp.module.volume_net.train()
    B = 8
    img, k2d, kc, gt = synth.synth_inputs(B, 256, 192, seed=12, with_gt=True)
    lo, hi = cd.shard_bounds(B, rank, world)
    kc_dev = kc[lo:hi].clone().to(device)
    pred = ddp(img[lo:hi].to(device), k2d[lo:hi].to(device), kc_dev)
    MPJPE()(pred, gt[lo:hi].to(device)).backward()                          # DDP averages the gradients inside backward (C3)
    torch.cuda.synchronize()
    inner = ddp.module
    got = {k: p.grad.detach().clone() for k, p in inner.named_parameters() if p.grad is not None}
    # the third argument is normalised IN PLACE through the wrapper as well (train.py:183 passes it un-cloned)
    ref = kc[lo:hi].clone()
    ref[..., 0] = ref[..., 0] / 96 - 1
    ref[..., 1] = ref[..., 1] / 128 - 1
    inplace_ok = torch.equal(kc_dev.cpu(), ref)
    # single-process gradient of the concatenated batch on the unwrapped module
    inner.zero_grad(set_to_none=True)
    pred = inner(img.to(device), k2d.to(device), kc.clone().to(device))
    MPJPE()(pred, gt.to(device)).backward()
    torch.cuda.synchronize()
    worst, n = 0.0, 0
    for k, p in inner.named_parameters():
        if not k.startswith("volume_net."):
            assert p.grad is None and k not in got                           # frozen backbone: no gradient, not in DDP's buckets
            continue
        rel = ((got[k] - p.grad).abs().max() / p.grad.abs().max().clamp_min(1e-12)).item()
        worst = max(worst, rel); n += 1
    same_weights = None
    w = inner.volume_net.head[1].weight.detach().cpu()
    ws = [torch.empty_like(w) for _ in range(world)]
    dist.all_gather(ws, w)
    same_weights = all(torch.equal(ws[0], t) for t in ws)
    keys = list(ddp.state_dict().keys())
    q.put((rank, path, n, worst, inplace_ok, same_weights, len(keys), keys[0]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sync_bn", [False, True], ids=["plain", "sync_bn_converted"])
def test_distributed_data_parallel_wrapper_step_equals_the_full_batch_gradient(sync_bn):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q, sync_bn)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, path, n, worst, inplace_ok, same_weights, nkeys, key0 in res:
        print(f"rank {rank}: DDP path = {path}; {n} gradients, worst |DDP-averaged shard grads - full-batch grad| / max = {worst:.2e}")
        assert n == 191 and worst <= 2e-5
        assert inplace_ok and same_weights
        assert nkeys == 1943 and key0.startswith("module.")
