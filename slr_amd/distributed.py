"""Multi-GPU driver: one process per GPU, image tiles dealt round-robin, ONE reduce at the end.

The reference has no distributed path (one process, a thread pool over 8x8 tiles:
PathTracingRenderer.cpp:72-81).  The tile is kept as the unit of distribution: rank r renders the
tiles t with t % world == r (slrhip_shard), writes zeros elsewhere, and a single sum-reduce of the float
framebuffer to rank 0 (RCCL over xGMI on GPUs; gloo in the CPU tests) assembles the image — supports are
disjoint, so the sum is a gather and the result does not depend on the world size.
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment; returns (rank, local_rank, world)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


def shard_for(rank, world):
    return (rank, world)


def reduce_framebuffer(fb, world, dst=0):
    """Sum the per-rank framebuffers (disjoint tile supports) onto rank `dst`; in place."""
    if world > 1:
        if fb.is_cuda and dist.get_backend() == "gloo":
            # gloo has no device-side reduce: host round trip (the CPU tests and bench.py's one-GPU rehearsal; RCCL never gets here)
            host = fb.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
            fb.copy_(host)
        else:
            dist.reduce(fb, dst=dst, op=dist.ReduceOp.SUM)
    return fb


def render_step(ctx, settings, spp, fb, rank, world, stream=None):
    """One bench step on this rank's GPU: shard render -> resolve into `fb` (a CUDA tensor) -> reduce."""
    ctx.render_begin(settings, shard=shard_for(rank, world))
    ctx.render(0, spp, stream)
    ctx.resolve_into(fb.data_ptr(), fb.numel(), stream)
    return reduce_framebuffer(fb, world)
