#!/bin/bash
# Two-stage segmentation of the RGB-D pairs under a directory (default: the demo frame shipped with the tests).
#   experiments/scripts/demo_rgbd_add.sh <first-stage checkpoint> <crop checkpoint> [image dir] [gpu]
set -e
export PYTHONUNBUFFERED=True
python tools/test_images.py --gpu "${4:-0}" \
  --network seg_resnet34_8s_embedding \
  --cfg experiments/cfgs/seg_resnet34_8s_embedding_cosine_rgbd_add_tabletop.yml \
  --pretrained "$1" --pretrained_crop "$2" \
  --imgdir "${3:-tests/golden/demo}"
