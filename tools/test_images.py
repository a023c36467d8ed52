#!/usr/bin/env python
"""Demo driver — the build's counterpart of the reference's tools/test_images.py:138-225:
list RGB-D pairs, build the two networks, run test_sample per frame, save label PNGs.

    python tools/test_images.py --imgdir tests/golden/demo [--pretrained ckpt.pth --pretrained_crop crop.pth]
                                [--cfg experiments/cfgs/<experiment>.yml --network seg_resnet34_8s_embedding]

Without checkpoints the calibrated synthetic weights (synth.calibrated_state_dict) are used.
"""
import argparse
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from unseenobjectclustering_amd import io as uio, networks, synth  # noqa: E402
from unseenobjectclustering_amd.fcn.config import cfg, cfg_from_file, network_mode, uses_depth  # noqa: E402
from unseenobjectclustering_amd.fcn.test_dataset import test_sample  # noqa: E402


def load_weights(path):
    if path is None:
        if network_mode() != "RGBD_ADD":
            return None      # no calibrated synthetic weights for the other modalities: xavier init (SEG.py:77-85)
        return {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
    data = torch.load(path, map_location="cpu")
    return data["model"] if isinstance(data, dict) and "model" in data else data   # tools/test_net.py:110-112


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--imgdir", required=True)
    ap.add_argument("--color", default="*-color.png")
    ap.add_argument("--depth", default="*-depth.png")
    ap.add_argument("--pretrained", default=None)
    ap.add_argument("--pretrained_crop", default=None)
    ap.add_argument("--outdir", default=None)
    ap.add_argument("--cfg", dest="cfg_file", default=None, help="experiment yml (tools/test_images.py:47-49)")
    ap.add_argument("--network", dest="network_name", default=None,
                    help="seg_resnet34_8s_embedding | seg_resnet34_8s_embedding_early (:62-64)")
    args = ap.parse_args()
    if args.cfg_file is not None:
        cfg_from_file(args.cfg_file)                               # :146-147

    np.random.seed(cfg.RNG_SEED)                                   # tools/test_images.py:152-154
    cfg.gpu_id = args.gpu
    cfg.device = torch.device("cuda:%d" % args.gpu)
    colors = sorted(glob.glob(os.path.join(args.imgdir, args.color)))
    depths = sorted(glob.glob(os.path.join(args.imgdir, args.depth)))
    assert len(colors) == len(depths) and colors, "need matching colour/depth images"
    cam_file = os.path.join(args.imgdir, "camera_params.json")
    cam = json.load(open(cam_file)) if os.path.exists(cam_file) else dict(synth.DEMO_CAMERA)
    name = args.network_name or ("seg_resnet34_8s_embedding_early" if network_mode() == "RGBD_EARLY"
                                 else "seg_resnet34_8s_embedding")
    network = networks.__dict__[name](2, cfg.TRAIN.NUM_UNITS, load_weights(args.pretrained)).eval()            # :191-199
    network_crop = networks.__dict__[name](2, cfg.TRAIN.NUM_UNITS, load_weights(args.pretrained_crop)).eval()  # :201-209
    outdir = args.outdir or args.imgdir
    for fc, fd in zip(colors, depths):
        if network_mode() == "RGBD_ADD" or uses_depth():
            sample = uio.read_sample_raw(fc, fd, cam)  # uint8/uint16 upload, prep fused on the device
        else:
            sample = uio.read_sample(fc, None, cam)    # COLOR: no depth in the sample (:110,131)
        out_label, out_label_refined = test_sample(sample, network, network_crop)
        final = out_label_refined if out_label_refined is not None else out_label
        from PIL import Image
        name = os.path.join(outdir, os.path.basename(fc)[:-4] + "-label.png")
        Image.fromarray(final[0].numpy().astype(np.uint8)).save(name)
        print("save data to {}  ({} segments)".format(name, int(final.max())))


if __name__ == "__main__":
    main()
