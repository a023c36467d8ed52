"""Dataset factory — mirror of /root/reference/lib/datasets/factory.py:7-43 for the evaluation datasets
(`osd_object_test`, `ocid_object_test`).  `tabletop_object_*` is the synthetic TRAINING set: out of scope."""
from .ocid_object import OCIDObject
from .osd_object import OSDObject

__sets = {}
for split in ["test"]:
    __sets["osd_object_{}".format(split)] = (lambda split=split: OSDObject(split))
    __sets["ocid_object_{}".format(split)] = (lambda split=split: OCIDObject(split))


def get_dataset(name):
    """Get an imdb (image database) by name."""
    if name not in __sets:
        if name.startswith("tabletop_object_"):
            raise NotImplementedError("the synthetic training set (tabletop_object_*) is not part of the inference path")
        raise KeyError("Unknown dataset: {}".format(name))
    return __sets[name]()


def list_datasets():
    return __sets.keys()
