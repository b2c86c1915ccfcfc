"""One process per GPU over RCCL / xGMI (replaces the reference's single-process nn.DataParallel,
models/utils.py:93).

* Sampling shards by replication: every rank runs its own PC sampler on its own batch with its own
  Philox stream; there is NO collective on that path (the Langevin step size is a per-shard batch mean,
  SURVEY F10).
* Training is data-parallel with ONE exchange per step: the flat fp32 gradient buffer
  (backward.FlatParams.grad, 62.76 M floats = 251 MB for NCSN++ CIFAR-10) is summed with a single
  all-reduce; the 1/world factor is folded into d loss / d score by the loss head, so SUM is the mean.
  `torch.distributed` backend "nccl" IS RCCL on ROCm; CPU tests use "gloo".
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def broadcast_parameters(model, src=0):
    """Make every replica start from rank `src`'s parameters (one broadcast of the flat buffer when it exists)."""
    if world_size() == 1:
        return
    flat = getattr(model, "_flat_params", None)
    if flat is not None and flat.owns(model):
        dist.broadcast(flat.data, src=src)
        flat.touch()            # out-of-band write: packed weight copies (engine.WeightStore) must be rebuilt
    else:
        with torch.no_grad():
            for p in model.parameters():
                buf = p.detach().clone()
                dist.broadcast(buf, src=src)
                p.copy_(buf)    # bumps Tensor._version, which WeightStore.refresh watches
                if getattr(p, "_ssde_flat", None) is not None:
                    p._ssde_flat.touch()
    for b in model.buffers():
        dist.broadcast(b.data, src=src)


def allreduce_gradients(flat):
    """Sum the flat gradient buffer over ranks (callers pre-scale by 1/world)."""
    if world_size() > 1:
        dist.all_reduce(flat.grad)


def shard_batch(batch, rank=None, world=None):
    """This rank's slice of a global batch along dim 0."""
    world = world_size() if world is None else world
    rank = (dist.get_rank() if world > 1 else 0) if rank is None else rank
    per = batch.shape[0] // world
    return batch[rank * per:(rank + 1) * per]
