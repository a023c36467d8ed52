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


def _bench(tmp_path, tag, force, steps=4):
    dump = os.path.join(str(tmp_path), tag + ".npy")
    env = dict(os.environ, UOC_BENCH_DUMP=dump, MASTER_PORT="29541")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    if force:
        env["UOC_BENCH_FORCE_DIST"] = "1"
    else:
        env.pop("UOC_BENCH_FORCE_DIST", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(steps), "--warmup", "1",
                        "--cpu-frames", "0", "--profile-steps", "0", "--sustained-seconds", "0"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1 and len(line[0]) < 3000
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


def test_a_rank_sized_block_through_the_collective(tmp_path, device):
    """BASELINE configs[4]: one rank of the 8-GPU job holds 1024 / 8 = 128 frames = a 39 MB uint8 label-map block.  The
    REAL frame function over 128 frames on this GPU with the nccl process group forced on (world size 1): the error-flag
    all_reduce + all_gather_into_tensor of that block is timed (`gather_s` of the rank, on the compact line too) and the
    line stays under the driver's tail."""
    coll, maps = _bench(tmp_path, "rank_block", True, steps=128)
    assert maps.shape == (128, 480, 640) and int(maps.max()) >= 5
    assert coll["config"]["collective"] is True and coll["config"]["total_frames"] == 128
    r = coll["per_rank"][0]
    assert r["frames"] == 128 and r["gather_s"] >= 0.0 and r["compute_s"] > 0.1
    assert r["gather_s"] < 0.25 * r["compute_s"], "the gather of one rank's block must stay a small part of the job"
    rec = {"frames": 128, "block_bytes": int(maps.nbytes), "compute_s": r["compute_s"], "gather_s": r["gather_s"],
           "frames_per_s": coll["value"], "world": 1, "backend": "nccl (RCCL), forced on one GPU"}
    json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "forced_dist_rank_block.json"), "w"))
    print(json.dumps(rec))


def test_eight_ranks_on_one_gpu_match_the_single_rank_block(tmp_path, device):
    """VERDICT r4 item 2a — the launch shape of the first real 8-GPU run with everything but RCCL in it: `python bench.py
    --gpus 8 --frames 64` spawns eight ranks (torch.distributed.run) that all run the REAL frame function on GPU 0
    (UOC_BENCH_ONE_DEVICE=1, gloo): rank 0 tunes and writes the tile cache while the others wait at the barrier and then load
    it, every rank runs its 8-frame block on three streams, the error flag and the configuration fingerprint are all-reduced,
    eight uint8 blocks are gathered.  The gathered [64, 480, 640] block must equal the single-process one, the line must hold
    8 per_rank entries (with the host CPU seconds each rank's Python loop burnt: DESIGN section 6's core budget) and stay
    under the driver's tail."""
    def run(tag, gpus, extra_env):
        dump = os.path.join(str(tmp_path), tag + ".npy")
        env = dict(os.environ, UOC_BENCH_DUMP=dump, **extra_env)
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "UOC_BENCH_FORCE_DIST"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--frames", "64", "--warmup", "1",
                            "--cpu-frames", "0", "--profile-steps", "0", "--sustained-seconds", "0", "--skip-pcie", "--skip-latency"],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(line) == 1 and len(line[0]) < 3000
        return json.loads(line[0]), np.load(dump)
    eight, maps8 = run("eight", 8, {"UOC_BENCH_ONE_DEVICE": "1", "UOC_BENCH_BACKEND": "gloo"})
    one, maps1 = run("one", 1, {})
    assert eight["n_gpus"] == 8 and eight["scaling"] == "strong" and eight["config"]["collective"] is True
    assert len(eight["per_rank"]) == 8 and all(r["frames"] == 8 for r in eight["per_rank"])
    assert maps1.shape == maps8.shape == (64, 480, 640)
    assert np.array_equal(maps1, maps8), "the gathered block of eight ranks differs from the single-rank block"
    json.dump({"world": 8, "frames": 64, "per_rank": eight["per_rank"], "frames_per_s_eight_ranks_one_gpu": eight["value"],
               "frames_per_s_one_rank": one["value"], "one_rank": one["per_rank"]},
              open(os.path.join(ROOT, "gpurun_out", "eight_ranks_one_gpu.json"), "w"))


def test_a_multi_rank_record_carries_roofline_cpu_baseline_and_parity(tmp_path, device):
    """at world > 1 rank 0 still fills the three record blocks north_star wants "in the same run" — the profiled
    pass (roofline + kernel shares), the CPU baseline and the parity of its first frames — after the gather, while the other ranks
    are done.  Two ranks on GPU 0 (UOC_BENCH_ONE_DEVICE=1, gloo)."""
    env = dict(os.environ, UOC_BENCH_ONE_DEVICE="1", UOC_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "UOC_BENCH_FORCE_DIST"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "16", "--warmup", "1",
                        "--cpu-frames", "1", "--profile-steps", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1 and len(line[0]) < 3000
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and len(d["per_rank"]) == 2
    # (one launch set on a GPU that two ranks share: WHICH kernel class leads and its numbers mean nothing here — rank 1's
    # teardown runs on the same device during rank 0's profiled pass; on its own GPU it is wino4_gemm, as in the N = 1 record)
    assert d["roofline"] and d["roofline"]["kernel"] and d["roofline"]["bound"] in ("mfma", "hbm")
    assert 0.0 < d["roofline"]["frac"] < 1.0 and d["roofline"]["avg_launch_us"] > 0 and d["roofline"]["peak"] > 0
    assert d["cpu_baseline"] and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port"
    assert d["parity"] and d["parity"]["frames"] == 1 and d["parity"]["embed_max_err"] < 1e-3
    assert d["kernel_time_share"] and d.get("kernel_time_share_pipe")
    assert d["latency"] is None and d["sustained_frames_per_s"] is None      # the N = 1 legs stay N = 1 legs
