#!/bin/bash
# ROS node on a RealSense-style camera (cfg TEST.ROS_CAMERA: D415); needs a ROS environment.
#   experiments/scripts/ros_seg_rgbd_add.sh <checkpoint> <crop checkpoint> [gpu]
set -e
export PYTHONUNBUFFERED=True
python ros/test_images_segmentation.py --gpu "${3:-0}" \
  --network seg_resnet34_8s_embedding \
  --cfg experiments/cfgs/seg_resnet34_8s_embedding_cosine_rgbd_add_tabletop.yml \
  --pretrained "$1" --pretrained_crop "$2"
