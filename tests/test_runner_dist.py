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


def _mismatch_worker(rank, world, port, out_dir, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if mode == "mismatch":          # rank 1 "runs another library configuration"
        runner.config_fingerprint = lambda: 1000 + (1 if rank == 1 else 0)
    else:                           # rank 1's library is missing / too old to have a fingerprint
        def broken():
            if rank == 1:
                raise AttributeError("undefined symbol: uoc_config_fingerprint")
            return 1000
        runner.config_fingerprint = broken
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=60))
    try:
        runner.run_sharded(4, _frame, 6, 8, torch.device("cpu"), rank, world)
        msg = "no error"
    except (RuntimeError, AttributeError) as e:
        msg = f"{type(e).__name__}: {e}"
    open(os.path.join(out_dir, f"r{rank}.txt"), "w").write(msg)
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_with_different_library_configurations_fail_before_the_gather(tmp_path):
    """one stray rounding-affecting knob (or another library build) on one rank would silently break
    sharding independence.  run_sharded all-reduces uoc_config_fingerprint() with its error flag; EVERY rank raises."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_mismatch_worker, args=(2, port, str(tmp_path), "mismatch"), nprocs=2, join=True)
    for r in range(2):
        assert "different libuoc_hip configurations" in open(os.path.join(str(tmp_path), f"r{r}.txt")).read()


def test_a_rank_without_a_fingerprint_fails_on_every_rank_instead_of_hanging(tmp_path):
    """config_fingerprint() raising on one rank (missing .so, a library that predates the symbol) used to
    leave the other ranks waiting in the all-reduce until the process-group timeout.  The rank now joins the all-reduce
    with an error flag: it re-raises its own error, the other rank raises 'another rank failed' — at once."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import time
    t0 = time.time()
    mp.spawn(_mismatch_worker, args=(2, port, str(tmp_path), "broken"), nprocs=2, join=True)
    assert time.time() - t0 < 45
    assert "another rank failed" in open(os.path.join(str(tmp_path), "r0.txt")).read()
    assert "uoc_config_fingerprint" in open(os.path.join(str(tmp_path), "r1.txt")).read()


def test_the_fingerprint_is_the_library_s():
    from unseenobjectclustering_amd import _native
    assert runner.config_fingerprint() == _native.config_fingerprint() & 0x7FFFFFFFFFFFFFFF > 0


def test_launch_set_plan():
    """The launch sets of a frame block: full sets, and a partial last round spread over all streams (speed only)."""
    f = runner.launch_set_sizes
    assert f(20, 4, 3) == [4, 4, 4, 3, 3, 2] and f(20, 4, 3, False) == [4, 4, 4, 4, 4]
    assert f(64, 4, 3) == [4] * 16                      # remainder = one set: untouched
    assert f(8, 4, 3) == [3, 3, 2] and f(5, 4, 3) == [2, 2, 1] and f(3, 4, 3) == [3] and f(12, 4, 3) == [4, 4, 4]
    assert f(7, 4, 1) == [4, 3] and f(0, 4, 3) == []
    for n in range(0, 70):
        for group in (1, 2, 4, 6):
            for depth in (1, 2, 3, 4):
                for tail in (False, True):
                    s = f(n, group, depth, tail)
                    assert sum(s) == n and all(1 <= v <= group for v in s)
