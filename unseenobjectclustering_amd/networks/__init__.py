"""Mirror of the reference's `networks` package surface for the hot path (lib/networks/SEG.py):
`networks.__dict__['seg_resnet34_8s_embedding'](num_classes, num_units, data)` builds the
ResNet34-8s embedding network for the configured modality, executed by libuoc_hip.so."""
from .SEG import SEGNET, seg_resnet34_8s_embedding, seg_resnet34_8s_embedding_early, update_model  # noqa: F401

__all__ = ["SEGNET", "seg_resnet34_8s_embedding", "seg_resnet34_8s_embedding_early", "update_model"]
