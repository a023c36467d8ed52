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
