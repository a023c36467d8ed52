#!/usr/bin/env python
"""Where does a headline frame's GPU label map differ from the oracle's, and by how much could it?

For global bench frame G (default 0) the script walks the two-stage path on both sides with the SAME first-seed draws
and compares every intermediate: stage-1 seeds / labels, ROI boxes, per-crop seed indices, seed labels and crop label
maps.  For every crop pixel whose cluster differs it reports, from the ORACLE's own embeddings and converged seeds, the
gap between the pixel's distance to the nearest seed of the oracle's cluster and to the nearest seed of the GPU's
cluster, next to the measured GPU-vs-oracle embedding difference at that pixel — a pixel can only flip when the gap is
of the order of the embedding error (|delta d| <= 0.5 * (|dx| + |dz|), cosine distance of unit vectors).

Runs on the GPU box:  python scripts/parity_analysis.py [G]  -> gpurun_out/parity_analysis_<G>.json
Imports oracle/ as the checker (test infrastructure), like tests/ do."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import backbone_oracle as BO, glue_oracle as GO, mean_shift_oracle as O  # noqa: E402
from unseenobjectclustering_amd import networks, runner, synth  # noqa: E402
from unseenobjectclustering_amd.fcn import test_dataset as TD  # noqa: E402
from unseenobjectclustering_amd.fcn.config import cfg  # noqa: E402
from unseenobjectclustering_amd.utils.mean_shift import cluster_batch  # noqa: E402


def main():
    g = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    dev = torch.device("cuda:0")
    cfg.device = dev
    torch.set_num_threads(min(64, len(os.sched_getaffinity(0))))
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
    net = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
    s = 10_000 + g
    fr = synth.palette_frame(s, 480, 640, 5 + s % 3)
    img, dep = torch.from_numpy(fr["image_color"]), torch.from_numpy(fr["depth"])
    rs = np.random.RandomState(runner.frame_rng_seed(g))
    first1 = rs.randint(0, 480 * 640)
    rep = {"frame": g}

    # ---- stage 1 ----
    f_cpu = BO.segnet_forward(sd, img, dep)
    X_cpu = f_cpu[0].reshape(64, -1).t().contiguous()
    lab_cpu, idx_cpu, parts = O.mean_shift_smart_init(X_cpu, 20.0, 100, 10, first_index=first1, epsilon=0.04, return_parts=True)
    f_gpu = net(img.to(dev), None, dep.to(dev))
    X_gpu = TD._pixel_major(f_gpu)
    lab_gpu, idx_gpu, Z_gpu, sl_gpu = cluster_batch(X_gpu, [first1], 20.0, 100, 10, 0.04, return_parts=True)
    rep["stage1"] = {
        "embedding_max_abs_diff": float((X_gpu[0].cpu() - X_cpu).abs().max()),
        "seed_indices_equal": bool(np.array_equal(idx_gpu[0].cpu().numpy(), idx_cpu.numpy())),
        "converged_seed_max_abs_diff": float((Z_gpu[0].cpu() - parts["Z"]).abs().max()),
        "seed_labels_equal": bool(np.array_equal(sl_gpu[0].cpu().numpy(), parts["seed_labels"].numpy())),
        "labels_equal_up_to_permutation": bool(O.labels_equal_up_to_permutation(lab_gpu[0].cpu().numpy(), lab_cpu.numpy())),
    }
    out_cpu = GO.filter_labels_depth(lab_cpu.view(1, 480, 640).float(), dep, 0.8)
    rgb_c, mask_c, rois, dep_c = GO.crop_rois(img, out_cpu.clone(), dep)
    out_gpu = TD.filter_labels_depth(lab_gpu.view(1, 480, 640).float().cpu(), dep.to(dev), 0.8)
    rgb_g, mask_g, rois_g, dep_g = TD.crop_rois(img.to(dev), out_gpu.clone(), dep.to(dev))
    K = rgb_c.shape[0]
    rep["rois"] = {"K": K, "boxes_equal": bool(np.array_equal(rois.numpy(), rois_g.cpu().numpy())),
                   "mask_crops_equal": bool(torch.equal(mask_c, mask_g.cpu())),
                   "rgb_crop_max_abs_diff": float((rgb_c - rgb_g.cpu()).abs().max()),
                   "xyz_crop_max_abs_diff": float((dep_c - dep_g.cpu()).abs().max())}

    # ---- stage 2 ----
    firsts = [rs.randint(0, 224 * 224) for _ in range(K)]
    f2_cpu = BO.segnet_forward(sd, rgb_c, dep_c)
    f2_gpu = net(rgb_g, None, dep_g)
    X2_gpu = TD._pixel_major(f2_gpu)
    l2_gpu, i2_gpu, Z2_gpu, sl2_gpu = cluster_batch(X2_gpu, firsts, 20.0, 100, 10, 0.04, return_parts=True)
    crops = []
    for k in range(K):
        Xk = f2_cpu[k].reshape(64, -1).t().contiguous()
        lk, ik, pk = O.mean_shift_smart_init(Xk, 20.0, 100, 10, first_index=firsts[k], epsilon=0.04, return_parts=True)
        a, b = l2_gpu[k].cpu().numpy().astype(np.int64), lk.numpy()
        entry = {"crop": k, "embedding_max_abs_diff": float((X2_gpu[k].cpu() - Xk).abs().max()),
                 "seed_indices_equal": bool(np.array_equal(i2_gpu[k].cpu().numpy(), ik.numpy())),
                 "first_differing_seed_step": None,
                 "converged_seed_max_abs_diff": float((Z2_gpu[k].cpu() - pk["Z"]).abs().max()),
                 "seed_labels_equal": bool(np.array_equal(sl2_gpu[k].cpu().numpy(), pk["seed_labels"].numpy())),
                 "labels_equal_up_to_permutation": bool(O.labels_equal_up_to_permutation(a, b)), "flipped_pixels": []}
        if not entry["seed_indices_equal"]:
            d = np.nonzero(i2_gpu[k].cpu().numpy() != ik.numpy())[0]
            entry["first_differing_seed_step"] = int(d[0])
        if not entry["labels_equal_up_to_permutation"]:
            # map GPU ids onto oracle ids by majority, list the pixels that disagree
            kb = int(b.max()) + 1
            table = np.bincount(a * kb + b, minlength=(int(a.max()) + 1) * kb).reshape(-1, kb)
            a2b = table.argmax(axis=1)
            bad = np.nonzero(a2b[a] != b)[0]
            dist = 0.5 * (1 - torch.mm(Xk, pk["Z"].t()))            # oracle distances [n, m]
            sl = pk["seed_labels"].numpy()
            # the assign step's labels BEFORE the largest<->0 swap are seed_labels[closest]; undo the swap per pixel
            for p in bad[:50]:
                own = int(pk["closest"][p])
                row = dist[p].numpy()
                order = np.argsort(row)
                other = next((int(j) for j in order if sl[j] != sl[own]), None)
                entry["flipped_pixels"].append({
                    "pixel": int(p), "oracle_nearest_seed_dist": float(row[own]),
                    "nearest_seed_of_another_cluster_dist": float(row[other]) if other is not None else None,
                    "gap": float(row[other] - row[own]) if other is not None else None,
                    "embedding_abs_diff_at_pixel": float((X2_gpu[k, p].cpu() - Xk[p]).abs().max())})
            entry["flipped_pixel_count"] = int(bad.size)
        crops.append(entry)
    rep["stage2"] = crops
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", f"parity_analysis_{g}.json")
    json.dump(rep, open(path, "w"), indent=1)
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
