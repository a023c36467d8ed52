#!/usr/bin/env python
"""ROS node — the build's counterpart of the reference's ros/test_images_segmentation.py:47-285: subscribe to a
time-synchronised colour + registered-depth pair, keep the latest frame, segment it with the two-stage path
(fcn.test_dataset.test_sample on the MI355X) and publish the label maps and overlay images.

    python ros/test_images_segmentation.py --network seg_resnet34_8s_embedding --pretrained ckpt.pth \\
           [--pretrained_crop crop.pth] [--cfg experiments/cfgs/<experiment>.yml] [--gpu 0] [--rand]

Topics (same names and encodings as the reference): `seg_label`, `seg_label_refined` (mono8, header of the colour
frame), `seg_image`, `seg_image_refined` (rgb8).  The input topics follow cfg.TEST.ROS_CAMERA ('D415', 'Azure', or a
kinect-style namespace, :66-88); intrinsics come from the camera_info message (:91-96).

The node logic (`SegmentationNode`) talks to ROS only through the small `RosApi` bundle, so it runs — and is tested,
tests/test_ros_node.py — without a ROS installation; `load_ros()` binds the real rospy / message_filters / cv_bridge.
Not carried over: the SCALES_BASE != 1 rescale (:121-124, needs cv2; every shipped yml uses 1.0) raises
NotImplementedError, and the overlays are a plain 50 % palette blend without the reference's contour lines
(visualisation is outside the scope of this build; the label topics are what downstream nodes consume)."""
import argparse
import os
import sys
import threading
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from unseenobjectclustering_amd import io as uio, networks  # noqa: E402
from unseenobjectclustering_amd.fcn.config import cfg, cfg_from_file  # noqa: E402
from unseenobjectclustering_amd.fcn import test_dataset  # noqa: E402


def camera_topics(camera):
    """Input topics and frame names per camera type (:66-88)."""
    if camera == "D415":
        return dict(rgb="/camera/color/image_raw", depth="/camera/aligned_depth_to_color/image_raw",
                    info="/camera/color/camera_info", base_frame="measured/base_link",
                    camera_frame="measured/camera_color_optical_frame")
    if camera == "Azure":
        return dict(rgb="/k4a/rgb/image_raw", depth="/k4a/depth_to_rgb/image_raw", info="/k4a/rgb/camera_info",
                    base_frame="measured/base_link", camera_frame="rgb_camera_link")
    frame = "%s_rgb_optical_frame" % camera
    return dict(rgb="/%s/rgb/image_color" % camera, depth="/%s/depth_registered/image" % camera,
                info="/%s/rgb/camera_info" % camera, base_frame=frame, camera_frame=frame)


def load_ros():
    """The real ROS bindings as one bundle (import errors surface here, not at module import)."""
    import message_filters
    import rospy
    from cv_bridge import CvBridge
    from sensor_msgs.msg import CameraInfo, Image
    return SimpleNamespace(rospy=rospy, message_filters=message_filters, CvBridge=CvBridge, Image=Image,
                           CameraInfo=CameraInfo)


PALETTE = np.array([[0, 0, 0]] + [[(37 * i + 60) % 256, (91 * i + 30) % 256, (149 * i + 90) % 256] for i in range(1, 256)],
                   dtype=np.float32)


def overlay(im_rgb_u8, label):
    """50 % blend of the frame with a fixed per-id colour (background untouched)."""
    lab = label.astype(np.int64) % 256
    out = im_rgb_u8.astype(np.float32)
    fg = lab > 0
    out[fg] = 0.5 * out[fg] + 0.5 * PALETTE[lab[fg]]
    return np.ascontiguousarray(out.round().astype(np.uint8))


class SegmentationNode:
    """Latest-frame mailbox + one segmentation per spin_once().  The subscriber callback and the segmentation loop run
    on different threads in ROS; the mailbox is the only shared state and is guarded by a lock (:32,127-131,137-144)."""

    def __init__(self, network, network_crop, ros, segment=None):
        self.network, self.network_crop, self.ros = network, network_crop, ros
        self.segment = segment or test_dataset.test_sample
        self.bridge = ros.CvBridge()
        self._lock = threading.Lock()
        self._frame = None            # (bgr uint8 [H,W,3], depth metres float32 [H,W], frame_id, stamp)
        rospy, mf = ros.rospy, ros.message_filters
        rospy.init_node("seg_rgb")
        self.pub = {name: rospy.Publisher(name, ros.Image, queue_size=10)
                    for name in ("seg_label", "seg_label_refined", "seg_image", "seg_image_refined", "seg_feature")}
        topics = camera_topics(cfg.TEST.ROS_CAMERA)
        self.base_frame = self.target_frame = topics["base_frame"]
        self.camera_frame = topics["camera_frame"]
        rgb_sub = mf.Subscriber(topics["rgb"], ros.Image, queue_size=10)
        depth_sub = mf.Subscriber(topics["depth"], ros.Image, queue_size=10)
        K = np.array(rospy.wait_for_message(topics["info"], ros.CameraInfo).K).reshape(3, 3)      # :91-96
        # Python floats: the float32 pixel grid then stays float32 under NumPy 2's promotion rules as it did under the
        # value-based casting the reference was written for
        self.fx, self.fy, self.px, self.py = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
        print(K)
        sync = mf.ApproximateTimeSynchronizer([rgb_sub, depth_sub], 1, 0.1)                       # :98-101
        sync.registerCallback(self.on_rgbd)

    # ---- subscriber side ---------------------------------------------------------------------------
    def decode_depth(self, msg):
        """Depth message -> float32 metres, or None for an encoding the node does not take (:105-114)."""
        if msg.encoding == "32FC1":
            return np.asarray(self.bridge.imgmsg_to_cv2(msg))
        if msg.encoding == "16UC1":
            return np.asarray(self.bridge.imgmsg_to_cv2(msg)).astype(np.float32) / 1000.0
        self.ros.rospy.logerr_throttle(1, "Unsupported depth type. Expected 16UC1 or 32FC1, got {}".format(msg.encoding))
        return None

    def on_rgbd(self, rgb, depth):
        depth_m = self.decode_depth(depth)
        if depth_m is None:
            return
        im = np.asarray(self.bridge.imgmsg_to_cv2(rgb, "bgr8"))
        if cfg.TEST.SCALES_BASE[0] != 1:
            raise NotImplementedError("TEST.SCALES_BASE[0] != 1: the rescaled input path is not part of this build")
        with self._lock:
            self._frame = (im.copy(), depth_m.copy(), rgb.header.frame_id, rgb.header.stamp)

    # ---- segmentation side -------------------------------------------------------------------------
    def take_frame(self):
        with self._lock:
            if self._frame is None:
                return None
            im, depth_m, frame_id, stamp = self._frame
            return im.copy(), depth_m.copy(), frame_id, stamp

    def _publish(self, topic, array, encoding, frame_id, stamp):
        msg = self.bridge.cv2_to_imgmsg(array, encoding) if encoding == "rgb8" else self.bridge.cv2_to_imgmsg(array)
        msg.header.stamp, msg.header.frame_id = stamp, frame_id
        if encoding != "rgb8":
            msg.encoding = encoding
        self.pub[topic].publish(msg)

    def spin_once(self):
        """Segments the latest frame and publishes the results; False when no frame has arrived yet (:133-204)."""
        got = self.take_frame()
        if got is None:
            return False
        im, depth_m, frame_id, stamp = got
        print("===========================================")
        sample = uio.make_sample_metric(im.astype(np.float32), depth_m, self.fx, self.fy, self.px, self.py)   # :148-162
        out_label, out_label_refined = self.segment(sample, self.network, self.network_crop)
        label = out_label[0].cpu().numpy()
        self._publish("seg_label", label.astype(np.uint8), "mono8", frame_id, stamp)
        print("%d objects" % (len(np.unique(label)) - 1))
        rgb = im[:, :, ::-1]
        refined = None
        if out_label_refined is not None:
            refined = out_label_refined[0].cpu().numpy()
            self._publish("seg_label_refined", refined.astype(np.uint8), "mono8", frame_id, stamp)
        self._publish("seg_image", overlay(rgb, label), "rgb8", frame_id, stamp)
        if refined is not None:
            self._publish("seg_image_refined", overlay(rgb, refined), "rgb8", frame_id, stamp)
        return True

    def spin(self, idle_sleep=0.002):
        """Segments frames until ROS shuts down; sleeps briefly while no frame has arrived (the reference spins hot)."""
        import time
        while not self.ros.rospy.is_shutdown():
            if not self.spin_once():
                time.sleep(idle_sleep)


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Segment unseen objects in ROS RGB-D streams")
    p.add_argument("--gpu", dest="gpu_id", default=0, type=int)
    p.add_argument("--instance", dest="instance_id", default=0, type=int)
    p.add_argument("--pretrained", dest="pretrained", default=None, type=str)
    p.add_argument("--pretrained_crop", dest="pretrained_crop", default=None, type=str)
    p.add_argument("--cfg", dest="cfg_file", default=None, type=str)
    p.add_argument("--dataset", dest="dataset_name", default="shapenet_scene_train", type=str)
    p.add_argument("--rand", dest="randomize", action="store_true")
    p.add_argument("--network", dest="network_name", default="seg_resnet34_8s_embedding", type=str)
    p.add_argument("--background", dest="background_name", default=None, type=str)
    return p.parse_args(argv)


def build_networks(args):
    def load(path):
        data = torch.load(path, map_location="cpu")
        return data["model"] if isinstance(data, dict) and "model" in data else data
    if not args.pretrained:
        print("no pretrained network specified")                                    # :256-258
        sys.exit()
    factory = networks.__dict__[args.network_name]
    network = factory(2, cfg.TRAIN.NUM_UNITS, load(args.pretrained)).eval()
    network_crop = factory(2, cfg.TRAIN.NUM_UNITS, load(args.pretrained_crop)).eval() if args.pretrained_crop else None
    return network, network_crop


def main(argv=None, ros=None):
    args = parse_args(argv)
    print("Called with args:")
    print(args)
    if args.cfg_file is not None:
        cfg_from_file(args.cfg_file)
    if not args.randomize:
        np.random.seed(cfg.RNG_SEED)
    cfg.gpu_id = args.gpu_id
    cfg.device = torch.device("cuda:{:d}".format(cfg.gpu_id))
    cfg.instance_id = args.instance_id
    cfg.MODE = "TEST"
    cfg.TEST.VISUALIZE = False
    network, network_crop = build_networks(args)
    node = SegmentationNode(network, network_crop, ros or load_ros())
    node.spin()


if __name__ == "__main__":
    main()
