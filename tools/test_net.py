#!/usr/bin/env python
"""Dataset evaluation driver — the build's counterpart of the reference's tools/test_net.py:60-131: pick a dataset by
name, wrap it in a DataLoader (batch cfg.TEST.IMS_PER_BATCH, no shuffle), build the network(s) from checkpoints and
run test_segnet (per-frame segmentation + metrics + .mat per frame + averaged report).

    python tools/test_net.py --dataset ocid_object_test --network seg_resnet34_8s_embedding \\
           --pretrained ckpt.pth [--pretrained_crop ckpt_crop.pth] [--cfg experiments/cfgs/<experiment>.yml]
           [--data-root <dir holding OCID/ or OSD/>] [--output-dir <dir>]

Differences from the reference, all additive: --data-root / --output-dir (the reference hard-codes data/ and output/
under its checkout), and the DataLoader uses worker processes only when --workers > 0."""
import argparse
import os
import pprint
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.utils.data  # noqa: E402

from unseenobjectclustering_amd import datasets, networks  # noqa: E402
from unseenobjectclustering_amd.fcn.config import cfg, cfg_from_file, get_output_dir  # noqa: E402
from unseenobjectclustering_amd.fcn.test_dataset import test_segnet  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Evaluate the two-stage segmentation on a dataset")
    p.add_argument("--gpu", dest="gpu_id", default=0, type=int)
    p.add_argument("--pretrained", dest="pretrained", default=None, type=str)
    p.add_argument("--pretrained_crop", dest="pretrained_crop", default=None, type=str)
    p.add_argument("--cfg", dest="cfg_file", default=None, type=str)
    p.add_argument("--dataset", dest="dataset_name", default="ocid_object_test", type=str)
    p.add_argument("--rand", dest="randomize", action="store_true", help="do not fix the random seed")
    p.add_argument("--network", dest="network_name", default="seg_resnet34_8s_embedding", type=str)
    p.add_argument("--data-root", default=None, help="directory that holds OCID/ and/or OSD/ (default: <repo>/data)")
    p.add_argument("--output-dir", default=None, help="where the .mat files go (default: <repo>/output/<EXP_DIR>/<dataset>)")
    p.add_argument("--workers", default=0, type=int)
    return p.parse_args(argv)


def load_checkpoint(path):
    data = torch.load(path, map_location="cpu")
    return data["model"] if isinstance(data, dict) and "model" in data else data        # tools/test_net.py:110-112


def build_dataset(name, data_root):
    if data_root is None:
        return datasets.get_dataset(name)
    if name.startswith("ocid_object_"):
        return datasets.OCIDObject(name[len("ocid_object_"):], os.path.join(data_root, "OCID"))
    if name.startswith("osd_object_"):
        return datasets.OSDObject(name[len("osd_object_"):], os.path.join(data_root, "OSD"))
    return datasets.get_dataset(name)


def main(argv=None):
    args = parse_args(argv)
    print("Called with args:")
    print(args)
    if args.cfg_file is not None:
        cfg_from_file(args.cfg_file)                                    # :65-66
    if len(cfg.TEST.CLASSES) == 0:
        cfg.TEST.CLASSES = cfg.TRAIN.CLASSES                            # :68-69
    print("Using config:")
    pprint.pprint(dict(cfg))
    if not args.randomize:
        np.random.seed(cfg.RNG_SEED)                                    # :73-75
    cfg.gpu_id = args.gpu_id
    cfg.device = torch.device("cuda:{:d}".format(cfg.gpu_id))
    print("GPU device {:d}".format(args.gpu_id))
    cfg.MODE = "TEST"
    dataset = build_dataset(args.dataset_name, args.data_root)
    dataloader = torch.utils.data.DataLoader(dataset, batch_size=cfg.TEST.IMS_PER_BATCH, shuffle=False,
                                             num_workers=args.workers)
    print("Use dataset `{:s}` for training".format(dataset.name))
    if len(cfg.INTRINSICS) > 0:                                         # :98-101
        dataset._intrinsic_matrix = np.array(cfg.INTRINSICS).reshape(3, 3)
        print(dataset._intrinsic_matrix)
    output_dir = args.output_dir or get_output_dir(dataset, None)
    print("Output will be saved to `{:s}`".format(output_dir))
    os.makedirs(output_dir, exist_ok=True)

    if not args.pretrained:
        print("no pretrained network specified")                       # :114-116
        sys.exit()
    network_data = load_checkpoint(args.pretrained)
    print("=> using pre-trained network '{}'".format(args.pretrained))
    network = networks.__dict__[args.network_name](dataset.num_classes, cfg.TRAIN.NUM_UNITS, network_data).eval()
    network_crop = None
    if args.pretrained_crop:
        network_crop = networks.__dict__[args.network_name](dataset.num_classes, cfg.TRAIN.NUM_UNITS,
                                                            load_checkpoint(args.pretrained_crop)).eval()
    return test_segnet(dataloader, network, output_dir, network_crop)    # :131


if __name__ == "__main__":
    main()
