"""The RCCL leg of bench.py on the hardware at hand (one GPU): UOC_BENCH_FORCE_DIST=1 initialises the `nccl` process
group with world size 1 and sends the label-map block through the error-flag all_reduce + all_gather_into_tensor that
the multi-GPU run uses (runner.run_sharded).  The gathered block must equal the ungathered one."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(tmp_path, tag, force):
    dump = os.path.join(str(tmp_path), tag + ".npy")
    env = dict(os.environ, UOC_BENCH_DUMP=dump, MASTER_PORT="29541")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    if force:
        env["UOC_BENCH_FORCE_DIST"] = "1"
    else:
        env.pop("UOC_BENCH_FORCE_DIST", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1",
                        "--cpu-frames", "0", "--profile-steps", "0", "--sustained-seconds", "0"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    return json.loads(line[0]), np.load(dump)


def test_forced_collective_block_equals_plain_block(tmp_path, device):
    plain, a = _bench(tmp_path, "plain", False)
    coll, b = _bench(tmp_path, "coll", True)
    assert plain["config"]["collective"] is False and coll["config"]["collective"] is True
    assert a.shape == b.shape == (4, 480, 640) and a.dtype == np.uint8
    assert np.array_equal(a, b), "all-gathered label-map block differs from the local block"
    assert int(a.max()) >= 5, "frames must segment into their objects"
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"plain_fps": plain["value"], "forced_collective_fps": coll["value"]},
              open(os.path.join(ROOT, "gpurun_out", "forced_dist.json"), "w"))
