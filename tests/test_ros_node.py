"""The ROS node (ros/test_images_segmentation.py, counterpart of the reference's :47-204) driven through a fake ROS
bundle: topic wiring per camera type, intrinsics from camera_info, depth decoding, the latest-frame mailbox, the
network inputs it builds (bit-identical to the input preparation pinned to the reference in tests/golden/prep.npz) and
the published messages.  CPU only: the segmentation call is a stub here; tests/test_segnet_prep_gpu.py runs the node
with the real two-stage path."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from tests import fake_ros
from unseenobjectclustering_amd import io as uio
from unseenobjectclustering_amd.fcn.config import cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = [615.5, 0.0, 322.25, 0.0, 614.75, 241.5, 0.0, 0.0, 1.0]


def load_node_module():
    spec = importlib.util.spec_from_file_location("uoc_ros_node", os.path.join(ROOT, "ros", "test_images_segmentation.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture
def rosmod():
    saved = cfg.TEST.ROS_CAMERA, cfg.TEST.SCALES_BASE, cfg.INPUT
    cfg.TEST.SCALES_BASE = (1.0,)
    yield load_node_module()
    cfg.TEST.ROS_CAMERA, cfg.TEST.SCALES_BASE, cfg.INPUT = saved


def frame(seed=0, H=48, W=64):
    rng = np.random.default_rng(seed)
    im = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    dep = rng.integers(0, 3000, size=(H, W)).astype(np.uint16)
    return im, dep


def stub_segment(calls):
    def segment(sample, network, network_crop):
        calls.append(sample)
        z = sample["depth"][0, 2]
        label = (z > 1.0).float() + (z > 2.0).float()
        refined = None if network_crop is None else (label * 2)[None]
        return label[None], refined
    return segment


@pytest.mark.parametrize("camera,rgb,depth,info,base", [
    ("D415", "/camera/color/image_raw", "/camera/aligned_depth_to_color/image_raw", "/camera/color/camera_info", "measured/base_link"),
    ("Azure", "/k4a/rgb/image_raw", "/k4a/depth_to_rgb/image_raw", "/k4a/rgb/camera_info", "measured/base_link"),
    ("kinect1", "/kinect1/rgb/image_color", "/kinect1/depth_registered/image", "/kinect1/rgb/camera_info", "kinect1_rgb_optical_frame"),
])
def test_wiring_per_camera(rosmod, camera, rgb, depth, info, base):
    cfg.TEST.ROS_CAMERA = camera
    ros = fake_ros.make(K)
    node = rosmod.SegmentationNode("net", None, ros, segment=stub_segment([]))
    assert ros.state.node == "seg_rgb" and ros.state.waited == [info]
    sync = ros.state.synchronizers[0]
    assert [s.topic for s in sync.subs] == [rgb, depth] and sync.queue_size == 1 and sync.slop == 0.1
    assert sync.callback == node.on_rgbd
    assert node.base_frame == base == node.target_frame
    assert (node.fx, node.fy, node.px, node.py) == (615.5, 614.75, 322.25, 241.5)
    assert set(node.pub) == {"seg_label", "seg_label_refined", "seg_image", "seg_image_refined", "seg_feature"}


def test_frames_inputs_and_messages(rosmod):
    cfg.TEST.ROS_CAMERA, cfg.INPUT = "camera", "RGBD"
    ros = fake_ros.make(K)
    calls = []
    node = rosmod.SegmentationNode("net", "crop", ros, segment=stub_segment(calls))
    assert node.spin_once() is False and not calls                       # nothing received yet
    im, dep = frame(1)
    rgb_msg = fake_ros.Image(im, "bgr8", "cam_frame", stamp=123.5)
    # an unsupported depth encoding is reported and the frame dropped
    node.on_rgbd(rgb_msg, fake_ros.Image(dep.astype(np.int32), "32SC1"))
    assert len(ros.state.errors) == 1 and "32SC1" in ros.state.errors[0] and node.spin_once() is False
    # 16UC1 millimetres; a later frame replaces an unconsumed one
    node.on_rgbd(fake_ros.Image(frame(2)[0], "bgr8", "old", 1.0), fake_ros.Image(frame(2)[1], "16UC1"))
    node.on_rgbd(rgb_msg, fake_ros.Image(dep, "16UC1"))
    assert node.spin_once() is True and len(calls) == 1
    want = uio.make_sample(im, dep, dict(fx=K[0], fy=K[4], x_offset=K[2], y_offset=K[5]))
    assert calls[0]["image_color"].dtype == torch.float32 and torch.equal(calls[0]["image_color"], want["image_color"])
    assert calls[0]["depth"].dtype == torch.float32 and torch.equal(calls[0]["depth"], want["depth"])
    z = dep.astype(np.float32) / 1000.0
    label = (z > 1.0).astype(np.uint8) + (z > 2.0).astype(np.uint8)
    lab_msg, = node.pub["seg_label"].sent
    assert lab_msg.encoding == "mono8" and lab_msg.data.dtype == np.uint8 and np.array_equal(lab_msg.data, label)
    assert lab_msg.header.frame_id == "cam_frame" and lab_msg.header.stamp == 123.5
    ref_msg, = node.pub["seg_label_refined"].sent
    assert ref_msg.encoding == "mono8" and np.array_equal(ref_msg.data, 2 * label) and ref_msg.header.stamp == 123.5
    for topic, lab in (("seg_image", label), ("seg_image_refined", 2 * label)):
        msg, = node.pub[topic].sent
        assert msg.encoding == "rgb8" and msg.data.shape == im.shape and msg.header.frame_id == "cam_frame"
        assert np.array_equal(msg.data[lab == 0], im[:, :, ::-1][lab == 0])          # background untouched
        assert (msg.data[lab > 0] != im[:, :, ::-1][lab > 0]).any()
    # 32FC1 metres go through unchanged; without a crop network no refined topics
    node2 = rosmod.SegmentationNode("net", None, fake_ros.make(K), segment=stub_segment(calls))
    node2.on_rgbd(rgb_msg, fake_ros.Image(z, "32FC1"))
    assert node2.spin_once() is True and torch.equal(calls[1]["depth"], want["depth"])
    assert not node2.pub["seg_label_refined"].sent and not node2.pub["seg_image_refined"].sent
    assert len(node2.pub["seg_label"].sent) == 1 and len(node2.pub["seg_image"].sent) == 1


def test_color_only_and_rescale_guard(rosmod):
    cfg.TEST.ROS_CAMERA, cfg.INPUT = "camera", "COLOR"
    calls = []
    node = rosmod.SegmentationNode("net", None, fake_ros.make(K),
                                   segment=lambda s, n, c: (calls.append(s) or torch.zeros(1, 48, 64), None))
    im, dep = frame(3)
    node.on_rgbd(fake_ros.Image(im, "bgr8"), fake_ros.Image(dep, "16UC1"))
    assert node.spin_once() and "depth" not in calls[0]
    cfg.TEST.SCALES_BASE = (0.5,)
    with pytest.raises(NotImplementedError):
        node.on_rgbd(fake_ros.Image(im, "bgr8"), fake_ros.Image(dep, "16UC1"))


def test_spin_stops_on_shutdown_and_main_requires_checkpoint(rosmod):
    ros = fake_ros.make(K, shutdown_after=3)
    node = rosmod.SegmentationNode("net", None, ros, segment=stub_segment([]))
    node.spin()
    assert ros.state.spins == 4
    with pytest.raises(SystemExit):
        rosmod.main(["--network", "seg_resnet34_8s_embedding"], ros=ros)          # :256-258: no checkpoint, no node
