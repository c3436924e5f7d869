"""CPU, world_size 2 over gloo: the N > 1 path (tile sharding + one framebuffer reduce) is exercised with the
oracle standing in for the GPU shard render; the assembled image must equal the single-process image bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from helpers import load_golden, scene_from_golden
    from oracle import binding as ob
    from slr_amd import distributed
    r, _, w = distributed.init("gloo")
    assert (r, w) == (rank, world)
    g = load_golden("rgb_cornell_matte")
    st = ob.settings(int(g["width"]), int(g["height"]), int(g["seed"]))
    part, ctr = ob.load("oracle").scene(scene_from_golden(g)).render(st, int(g["spp"]), shard=distributed.shard_for(rank, world), threads=2)
    fb = torch.from_numpy(part.copy())
    distributed.reduce_framebuffer(fb, world)
    samples = torch.tensor([ctr.samples], dtype=torch.int64)
    torch.distributed.all_reduce(samples)
    if rank == 0:
        np.savez(out_path, fb=fb.numpy(), samples=int(samples.item()), want=g["framebuffer"],
                 expect_samples=int(g["width"]) * int(g["height"]) * int(g["spp"]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_render_assembles_to_the_full_image(tmp_path, world):
    out = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r = np.load(out)
    assert int(r["samples"]) == int(r["expect_samples"])             # every pixel rendered by exactly one rank
    assert (r["fb"].view(np.uint32) == r["want"].view(np.uint32)).all()
