"""A stand-in for the ROS bindings the segmentation node talks to (rospy, message_filters, cv_bridge, sensor_msgs):
just enough behaviour to drive ros/test_images_segmentation.py without a ROS installation.  Test infrastructure."""
from types import SimpleNamespace

import numpy as np


class Header:
    def __init__(self, frame_id="", stamp=None):
        self.frame_id, self.stamp = frame_id, stamp


class Image:
    def __init__(self, array=None, encoding="", frame_id="", stamp=None):
        self.data, self.encoding, self.header = array, encoding, Header(frame_id, stamp)


class CameraInfo:
    def __init__(self, K):
        self.K = list(K)


class CvBridge:
    def imgmsg_to_cv2(self, msg, desired_encoding="passthrough"):
        if desired_encoding == "bgr8" and msg.encoding == "rgb8":
            return msg.data[:, :, ::-1]
        return msg.data

    def cv2_to_imgmsg(self, array, encoding="passthrough"):
        if encoding == "passthrough":
            encoding = {np.dtype(np.uint8): "8UC1" if array.ndim == 2 else "8UC3"}[array.dtype]
        return Image(array, encoding)


class Publisher:
    def __init__(self, topic, msg_type, queue_size=None):
        self.topic, self.sent = topic, []

    def publish(self, msg):
        self.sent.append(msg)


class Subscriber:
    def __init__(self, topic, msg_type, queue_size=None):
        self.topic = topic


class ApproximateTimeSynchronizer:
    def __init__(self, subs, queue_size, slop):
        self.subs, self.queue_size, self.slop, self.callback = subs, queue_size, slop, None

    def registerCallback(self, cb):
        self.callback = cb


def make(K, shutdown_after=None):
    """A fresh fake ROS bundle; `state` records what the node did."""
    state = SimpleNamespace(node=None, waited=[], errors=[], synchronizers=[], spins=0)

    def init_node(name):
        state.node = name

    def wait_for_message(topic, msg_type):
        state.waited.append(topic)
        return CameraInfo(K)

    def is_shutdown():
        state.spins += 1
        return shutdown_after is not None and state.spins > shutdown_after

    class Sync(ApproximateTimeSynchronizer):
        def __init__(self, *a):
            super().__init__(*a)
            state.synchronizers.append(self)

    rospy = SimpleNamespace(init_node=init_node, Publisher=Publisher, wait_for_message=wait_for_message,
                            logerr_throttle=lambda period, text: state.errors.append(text), is_shutdown=is_shutdown)
    mf = SimpleNamespace(Subscriber=Subscriber, ApproximateTimeSynchronizer=Sync)
    return SimpleNamespace(rospy=rospy, message_filters=mf, CvBridge=CvBridge, Image=Image, CameraInfo=CameraInfo,
                           state=state)
