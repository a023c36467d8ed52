"""MI355X-native inference hot path of UnseenObjectClustering (RGB-D ResNet34-8s embedding +
seeded mean-shift + two-stage refinement) behind the reference's Python call surface.

    from unseenobjectclustering_amd.fcn.config import cfg
    from unseenobjectclustering_amd import networks
    from unseenobjectclustering_amd.fcn.test_dataset import test_sample, test_segnet
    from unseenobjectclustering_amd.utils.mean_shift import mean_shift_smart_init

All device work runs in libuoc_hip.so (hand-written HIP for gfx950, C ABI in include/uoc_hip.h).
"""
__version__ = "0.1.0"
