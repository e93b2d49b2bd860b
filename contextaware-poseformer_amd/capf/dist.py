"""Data-parallel helpers: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).  The hot path shards by frames and needs NO data-path collective
(SURVEY.md §8e); what does communicate in the reference is restated here:

  shard_bounds          validation sharding            ContextPose/mvn/datasets/human36m.py:536-552
  gather_predictions    padded all_gather of results   ContextPose/train.py:216-226          (C4)
  allreduce_mean_       DDP gradient averaging         ContextPose/train.py:361-362, :195    (C3)
  broadcast_state_      DDP parameter broadcast        ContextPose/train.py:362              (C1)
  max_over_ranks        bench.py timing rule
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """env:// rendezvous like train.py:240-249 (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(n, rank, world):
    """Contiguous n//world frames per rank, the remainder goes to the LAST rank (human36m.py:536-552)."""
    per = n // world
    lo = rank * per
    hi = n if rank == world - 1 else lo + per
    return lo, hi


def gather_predictions(local, total, world=None):
    """All ranks' [n_r, ...] tensors -> [total, ...] in rank order.  Shards are zero-padded to the
    largest (= last) shard, all_gather'ed and trimmed, as train.py:216-226 does."""
    if not dist.is_initialized():
        return local
    world = world or dist.get_world_size()
    per = total // world
    largest = total - per * (world - 1)
    pad = torch.zeros((largest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    parts = [b[: (per if r < world - 1 else largest)] for r, b in enumerate(bufs)]
    return torch.cat(parts, 0)


def allreduce_sum_(flat):
    """In-place SUM of one flat fp32 gradient buffer over ranks — one collective for all 14 M lifter gradients instead of
    DDP's three 25 MB buckets.  Returns (flat, scale) with scale = 1 / world_size: the division that turns the sum into DDP's
    average is folded into the consumer (capf_adamw_step's grad_scale), not run as a separate pass over 56 MB."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world > 1:
        if flat.is_cuda and dist.get_backend() == "gloo":        # CPU-staged (smoke tests of the N>1 flow on one GPU)
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat, 1.0 / world


def allreduce_mean_(flat):
    """allreduce_sum_ followed by the division (for consumers that want the averaged gradient itself, e.g. torch optimizers)."""
    flat, scale = allreduce_sum_(flat)
    if scale != 1.0:
        flat.mul_(scale)
    return flat


def broadcast_state_(module, src=0):
    """Rank `src`'s parameters and buffers to everyone, as one flat fp32 message per dtype."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    tensors = [t for t in module.state_dict().values() if t.is_floating_point()]
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.broadcast(flat, src)
    off = 0
    for t in tensors:
        t.copy_(flat[off: off + t.numel()].view_as(t))
        off += t.numel()


def max_over_ranks(seconds, device):
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return seconds
    if dist.get_backend() == "gloo":
        device = torch.device("cpu")
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
