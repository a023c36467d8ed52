"""Frame-parallel runner on CPU: world_size-2 gloo processes, stub per-frame function."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from unseenobjectclustering_amd import runner


def test_shard_range_covers_everything():
    for F in (1, 7, 8, 1024, 1025):
        for G in (1, 2, 3, 8):
            seen = []
            for r in range(G):
                lo, hi = runner.shard_range(F, r, G)
                assert 0 <= lo <= hi <= F
                seen += list(range(lo, hi))
            assert seen == list(range(F))


def _frame(i):
    """Deterministic stub 'segmentation' that consumes the per-frame RNG like the real path does."""
    first = np.random.randint(0, 1000)
    m = torch.full((6, 8), i % 250, dtype=torch.int32)
    m[0, 0] = first % 250
    return m


def _worker(rank, world, port, F, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = runner.run_sharded(F, _frame, 6, 8, torch.device("cpu"), rank, world)
    torch.save(full, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("F", [5, 8])
def test_two_rank_gather_equals_single_process(tmp_path, F):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, F, str(tmp_path)), nprocs=2, join=True)
    single = runner.run_sharded(F, _frame, 6, 8, torch.device("cpu"), 0, 1)
    for r in range(2):
        got = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))
        assert got.shape == (F, 6, 8) and torch.equal(got, single)
