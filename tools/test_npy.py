#!/usr/bin/env python
"""Driver for frames packed as .npy dictionaries — the build's counterpart of the reference's tools/test_npy.py
(`read_sample` :105-144, main loop :147-240): every *.npy under --imgdir is one RGB-D frame, segmented with the
two-stage path.

    python tools/test_npy.py --imgdir <dir> --network seg_resnet34_8s_embedding --pretrained ckpt.pth
                             [--pretrained_crop crop.pth] [--cfg experiments/cfgs/<experiment>.yml] [--outdir <dir>]

Two file layouts, as in the reference: {'rgb': uint8 RGB, 'depth': uint16 millimetres} with the intrinsics from
<imgdir>/camera_params.json, or {'debug_info': {'rgb', 'depth_image' (metres), 'intrinsics' (3x3)}}.
The reference's loop keeps no output (its save_data is never called); --outdir (additive) writes <name>-label.png."""
import argparse
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from unseenobjectclustering_amd import io as uio, networks  # noqa: E402
from unseenobjectclustering_amd.fcn.config import cfg, cfg_from_file  # noqa: E402
from unseenobjectclustering_amd.fcn.test_dataset import test_sample  # noqa: E402


def read_sample(filename, camera_params):
    """tools/test_npy.py:105-144: RGB -> BGR, /255 - PIXEL_MEANS/255; depth -> XYZ with the pinhole model.  Intrinsics
    enter the float32 pixel grid arithmetic as Python floats (float32 results under NumPy 1 and 2 alike) and a
    `depth_image` is taken as float32 metres."""
    d = np.load(filename, allow_pickle=True, encoding="latin1").item()
    if "debug_info" in d:
        info = d["debug_info"]
        K = np.asarray(info["intrinsics"])
        depth_m = np.asarray(info["depth_image"], dtype=np.float32)
        rgb = info["rgb"]
        fx, fy, px, py = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    else:
        depth_m = d["depth"].astype(np.float32) / 1000.0
        rgb = d["rgb"]
        fx, fy, px, py = (camera_params[k] for k in ("fx", "fy", "x_offset", "y_offset"))
    bgr = np.ascontiguousarray(np.asarray(rgb).astype(np.float32)[:, :, ::-1])
    return uio.make_sample_metric(bgr, depth_m, fx, fy, px, py)


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Segment RGB-D frames stored as .npy dictionaries")
    p.add_argument("--gpu", dest="gpu_id", default=0, type=int)
    p.add_argument("--pretrained", dest="pretrained", default=None, type=str)
    p.add_argument("--pretrained_crop", dest="pretrained_crop", default=None, type=str)
    p.add_argument("--cfg", dest="cfg_file", default=None, type=str)
    p.add_argument("--dataset", dest="dataset_name", default="shapenet_scene_train", type=str)
    p.add_argument("--imgdir", dest="imgdir", required=True, type=str)
    p.add_argument("--rand", dest="randomize", action="store_true")
    p.add_argument("--network", dest="network_name", default="seg_resnet34_8s_embedding", type=str)
    p.add_argument("--outdir", default=None, help="write <name>-label.png here (the reference keeps no output)")
    return p.parse_args(argv)


def main(argv=None, networks_override=None):
    args = parse_args(argv)
    print("Called with args:")
    print(args)
    if args.cfg_file is not None:
        cfg_from_file(args.cfg_file)
    if len(cfg.TEST.CLASSES) == 0:
        cfg.TEST.CLASSES = cfg.TRAIN.CLASSES
    if not args.randomize:
        np.random.seed(cfg.RNG_SEED)
    cfg.gpu_id = args.gpu_id
    cfg.device = torch.device("cuda:{:d}".format(cfg.gpu_id))
    cfg.instance_id = 0
    cfg.MODE = "TEST"
    images = sorted(glob.glob(os.path.join(args.imgdir, "*.npy")))
    cam_file = os.path.join(args.imgdir, "camera_params.json")
    camera_params = json.load(open(cam_file)) if os.path.exists(cam_file) else None
    if networks_override is not None:
        network, network_crop = networks_override
    else:
        if not args.pretrained:
            print("no pretrained network specified")          # :196-198
            sys.exit()
        def load(path):
            data = torch.load(path, map_location="cpu")
            return data["model"] if isinstance(data, dict) and "model" in data else data       # tools/test_net.py:110-112
        factory = networks.__dict__[args.network_name]
        network = factory(2, cfg.TRAIN.NUM_UNITS, load(args.pretrained)).eval()
        network_crop = factory(2, cfg.TRAIN.NUM_UNITS, load(args.pretrained_crop)).eval() if args.pretrained_crop else None
    order = np.random.permutation(len(images)) if cfg.TEST.VISUALIZE else range(len(images))      # :218-221
    results = []
    for i in order:
        sample = read_sample(images[i], camera_params)
        out_label, out_label_refined = test_sample(sample, network, network_crop)
        results.append((images[i], out_label, out_label_refined))
        if args.outdir:
            from PIL import Image
            os.makedirs(args.outdir, exist_ok=True)
            final = out_label_refined if out_label_refined is not None else out_label
            name = os.path.join(args.outdir, os.path.basename(images[i])[:-4] + "-label.png")
            Image.fromarray(final[0].numpy().astype(np.uint8)).save(name)
            print("save data to {}".format(name))
    return results


if __name__ == "__main__":
    main()
