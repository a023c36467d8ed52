"""Is the ORACLE's (= the reference's torch-CPU arithmetic's) label map reproducible across intra-op thread counts?
Runs oracle/glue_oracle.test_sample on bench frames with 1 / 4 / 16 / 64 threads and compares the final maps with the
4-thread run and with the committed fixture (tests/golden/bench_oracle, generated with 4 threads in the build container).
Test infrastructure (imports oracle/).  Usage: python scripts/oracle_thread_sensitivity.py [frames...]"""
import glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import backbone_oracle as BO, glue_oracle as GO
from unseenobjectclustering_amd import runner, synth

frames = [int(a) for a in sys.argv[1:]] or [0, 1]
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
net = lambda img, label, depth: BO.segnet_forward(sd, img, depth)
fix = {}
for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "bench_oracle", "frames_*.npz"))):
    z = np.load(path)
    fin = z["final"]
    for i in range(len(fin)):
        fix[int(z["first"]) + i] = fin[i]
out = []
for g in frames:
    s = 10_000 + g
    fr = synth.palette_frame(s, 480, 640, 5 + s % 3)
    img, dep = torch.from_numpy(fr["image_color"]), torch.from_numpy(fr["depth"])
    maps = {}
    for th in (4, 1, 16, 64):
        torch.set_num_threads(th)
        o, r = GO.test_sample(img, dep, net, net, np.random.RandomState(runner.frame_rng_seed(g)))
        maps[th] = (r if r is not None else o)[0].numpy().astype(np.int64)
    row = {"frame": g, "cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t")}
    for th in (1, 16, 64):
        row[f"pixels_differing_{th}_vs_4_threads"] = int((maps[th] != maps[4]).sum())
    if g in fix:
        row["pixels_differing_4_threads_vs_fixture"] = int((maps[4] != fix[g]).sum())
        row["pixels_differing_64_threads_vs_fixture"] = int((maps[64] != fix[g]).sum())
    print(json.dumps(row), flush=True)
    out.append(row)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "oracle_thread_sensitivity.json"), "w"), indent=1)
