#!/bin/bash
# Frames packed as .npy dictionaries.
#   experiments/scripts/test_npy.sh <checkpoint> <crop checkpoint> <dir with *.npy> [output dir]
set -e
export PYTHONUNBUFFERED=True
python tools/test_npy.py --network seg_resnet34_8s_embedding \
  --cfg experiments/cfgs/seg_resnet34_8s_embedding_cosine_rgbd_add_tabletop.yml \
  --pretrained "$1" --pretrained_crop "$2" --imgdir "$3" ${4:+--outdir "$4"}
