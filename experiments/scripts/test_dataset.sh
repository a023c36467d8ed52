#!/bin/bash
# Dataset evaluation (OCID or OSD) for one input modality.
#   experiments/scripts/test_dataset.sh <ocid|osd> <rgbd_add|rgbd_early|rgbd_cat|color|depth> <checkpoint> <crop checkpoint> [data root] [gpu]
set -e
export PYTHONUNBUFFERED=True
NET=seg_resnet34_8s_embedding; [ "$2" = rgbd_early ] && NET=seg_resnet34_8s_embedding_early
python tools/test_net.py --gpu "${6:-0}" \
  --network $NET \
  --cfg experiments/cfgs/seg_resnet34_8s_embedding_cosine_$2_tabletop.yml \
  --dataset "$1"_object_test \
  --pretrained "$3" --pretrained_crop "$4" ${5:+--data-root "$5"}
