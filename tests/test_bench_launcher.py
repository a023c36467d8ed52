"""bench.py's own launcher and sharding plumbing on CPU: `python bench.py --gpus 2` without torchrun must spawn its two
ranks (gloo, stub frame function) and the gathered label-map block must equal the single-process one."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, tag, *flags):
    dump = os.path.join(str(tmp_path), tag + ".npy")
    env = dict(os.environ, UOC_BENCH_DUMP=dump)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", *flags], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line on stdout (rank 0)"
    assert len(lines[0]) < 3000, "the driver keeps a few KB of output tail: the line must stay compact"
    return json.loads(lines[0]), np.load(dump)


@pytest.mark.parametrize("mode", [("--frames", "7"), ("--steps", "3")])
def test_self_launch_two_ranks_matches_single_process(tmp_path, mode):
    two, maps2 = _run(tmp_path, "two", "--gpus", "2", "--warmup", "1", *mode)
    one_flags = mode if mode[0] == "--frames" else ("--steps", "6")       # weak: 2 ranks x 3 frames = frames 0..5
    one, maps1 = _run(tmp_path, "one", "--gpus", "1", "--warmup", "1", *one_flags)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["scaling"] == ("strong" if mode[0] == "--frames" else "weak")
    assert two["config"]["total_frames"] == one["config"]["total_frames"] == maps1.shape[0]
    assert two["config"]["collective"] is True
    assert np.array_equal(maps1, maps2), "gathered block depends on the sharding"
    for key in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype",
                "data", "roofline", "cpu_baseline", "parity", "sustained_frames_per_s", "pcie_inclusive_frames_per_s", "per_rank",
                "full"):
        assert key in two
    assert len(two["per_rank"]) == 2 and all("gather_s" in r and "compute_s" in r for r in two["per_rank"])


def test_wrong_world_size_is_refused(tmp_path):
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--gpus", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_compact_line_of_a_full_record_stays_small():
    """The worst case seen so far (round 3's 22.7 KB record, 8 ranks) through bench.compact_line."""
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r03_bench.json")))
    full["per_rank"] = [dict(full["per_rank"][0], gather_s=0.0123) for _ in range(8)]
    full["n_gpus"] = 8
    line = json.dumps(bench.compact_line(full, "gpurun_out/bench_full.json"), separators=(",", ":"))
    assert len(line) < bench.MAX_LINE_BYTES
    d = json.loads(line)
    assert d["roofline"]["frac"] == full["roofline"]["frac"] and d["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]
    assert "by_shape" not in d["roofline"] and "conv_by_shape" not in d and "kernels" not in d


def test_self_launch_eight_ranks_matches_single_process(tmp_path):
    """The launch shape of BASELINE configs[4] (8 ranks, strong scaling) through the whole plumbing on CPU: self-launcher,
    tune-cache barriers (the two barriers pair up on every rank also in --stub runs), error-flag + configuration-fingerprint
    all-reduce, all_gather of 8 blocks, 8 per_rank entries on a line that stays under the driver's tail."""
    eight, maps8 = _run(tmp_path, "eight", "--gpus", "8", "--warmup", "1", "--frames", "19")     # ragged: blocks of 3, last rank short
    one, maps1 = _run(tmp_path, "one", "--gpus", "1", "--warmup", "1", "--frames", "19")
    assert eight["n_gpus"] == 8 and eight["scaling"] == "strong" and eight["config"]["collective"] is True
    assert len(eight["per_rank"]) == 8 and sum(r["frames"] for r in eight["per_rank"]) == 19
    assert all("host_cpu_s" in r for r in eight["per_rank"])
    assert maps8.shape == maps1.shape == (19, 12, 16) and np.array_equal(maps1, maps8)


def test_compact_line_never_exceeds_the_cap_whatever_the_record_holds():
    """the optional parts are dropped in a loop that re-measures; per-frame parity lists are summarised; a
    record without config.workload does not raise."""
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_full.json")))
    full["parity"]["mismatched_pixels"] = list(range(400))                       # a long --cpu-frames run
    full["parity"]["given_oracle_embeddings_mismatched_pixels"] = [0] * 400
    full["config"]["workload"] = "x" * 5000
    full["cpu_baseline"]["sample"] = "y" * 5000
    full["per_rank"] = [dict(full["per_rank"][0], gather_s=0.0123, host_cpu_s=1.234567) for _ in range(8)]
    line = json.dumps(bench.compact_line(full, "gpurun_out/bench_full.json"), separators=(",", ":"))
    assert len(line) <= bench.MAX_LINE_BYTES
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "full"):
        assert key in d
    assert d["parity"]["mismatched_pixels"]["frames"] == 400 and d["parity"]["mismatched_pixels"]["max"] == 399
    del full["config"]["workload"]
    assert len(json.dumps(bench.compact_line(full, "p"), separators=(",", ":"))) <= bench.MAX_LINE_BYTES
    fr = bench.frame_roofline(7.1, 5.97e-3)
    assert fr["mfma_frac"] < 1.0 < fr["algorithmic_mfma_frac"] and abs(fr["executed_gflop"] - (fr["algorithmic_gflop"] - 0.75 * (416.2 + 68.0 * 7.1))) < 0.2


def _run_fault(fault, *flags, launcher="self", timeout=150):
    """bench.py --stub with one rank misbehaving (UOC_BENCH_FAULT, honoured only with --stub).  launcher 'self': bench.py spawns
    its ranks; 'torchrun': the driver's form (python -m torch.distributed.run ... bench.py --gpus N ...)."""
    import socket
    env = dict(os.environ, UOC_BENCH_FAULT=fault, UOC_BENCH_TIMEOUT="90")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    base = [os.path.join(ROOT, "bench.py"), "--stub", "--gpus", "2", "--steps", "3", "--warmup", "1", "--dist-timeout", "12", *flags]
    if launcher == "torchrun":
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + base
    else:
        cmd = [sys.executable] + base
    import time
    t0 = time.time()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    took = time.time() - t0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    return r, lines, took


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
@pytest.mark.parametrize("fault,stages", [("1:setup:raise", ("setup", "timed", "gather")), ("1:init:hang", ("init",)),
                                           ("1:timed:exit", ("timed", "gather", "setup")), ("0:setup:raise", ("setup",))])
def test_a_failing_rank_leaves_one_error_record(fault, stages, launcher):
    """the first multi-GPU run must not be able to fail silently.  A rank that raises during set-up, a
    rank that never joins the rendezvous, a rank that dies inside the timed region — and rank 0 itself raising — each end
    in a non-zero exit code and exactly ONE JSON line {"metric", "value": null, "error", "stage", "n_gpus"} within a bounded
    time, under bench.py's own launcher and in the driver's torch.distributed.run form."""
    r, lines, took = _run_fault(fault, launcher=launcher)
    assert r.returncode != 0
    assert len(lines) == 1, (r.stdout[-1500:], r.stderr[-1500:])
    rec = json.loads(lines[0])
    assert rec["value"] is None and rec["n_gpus"] == 2 and rec["metric"].startswith("frames/sec")
    assert rec["error"] and rec["stage"] in stages + ("launch",), rec
    assert took < 120, f"the error record took {took:.0f} s"
