#!/usr/bin/env python
"""Demo driver — the build's counterpart of the reference's tools/test_images.py:138-225:
list RGB-D pairs, build the two networks, run test_sample per frame, save label PNGs.

    python tools/test_images.py --imgdir tests/golden/demo [--pretrained ckpt.pth --pretrained_crop crop.pth]

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
from unseenobjectclustering_amd.fcn.config import cfg  # noqa: E402
from unseenobjectclustering_amd.fcn.test_dataset import test_sample  # noqa: E402


def load_weights(path):
    if path is None:
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
    args = ap.parse_args()

    np.random.seed(cfg.RNG_SEED)                                   # tools/test_images.py:152-154
    cfg.gpu_id = args.gpu
    cfg.device = torch.device("cuda:%d" % args.gpu)
    colors = sorted(glob.glob(os.path.join(args.imgdir, args.color)))
    depths = sorted(glob.glob(os.path.join(args.imgdir, args.depth)))
    assert len(colors) == len(depths) and colors, "need matching colour/depth images"
    cam_file = os.path.join(args.imgdir, "camera_params.json")
    cam = json.load(open(cam_file)) if os.path.exists(cam_file) else dict(synth.DEMO_CAMERA)
    network = networks.seg_resnet34_8s_embedding(2, cfg.TRAIN.NUM_UNITS, load_weights(args.pretrained)).eval()
    network_crop = networks.seg_resnet34_8s_embedding(2, cfg.TRAIN.NUM_UNITS, load_weights(args.pretrained_crop)).eval()
    outdir = args.outdir or args.imgdir
    for fc, fd in zip(colors, depths):
        sample = uio.read_sample_raw(fc, fd, cam)      # uint8/uint16 upload, prep fused on the device
        out_label, out_label_refined = test_sample(sample, network, network_crop)
        final = out_label_refined if out_label_refined is not None else out_label
        from PIL import Image
        name = os.path.join(outdir, os.path.basename(fc)[:-4] + "-label.png")
        Image.fromarray(final[0].numpy().astype(np.uint8)).save(name)
        print("save data to {}  ({} segments)".format(name, int(final.max())))


if __name__ == "__main__":
    main()
