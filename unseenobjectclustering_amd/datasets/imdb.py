"""Image-database base class — the part of /root/reference/lib/datasets/imdb.py:13-33 the evaluation path reads
(`name`, `num_classes`, `classes`; tools/test_net.py:95,114, lib/fcn/test_dataset.py:299-305)."""


class imdb(object):
    def __init__(self):
        self._name = ""
        self._classes = []
        self._class_colors = []

    @property
    def name(self):
        return self._name

    @property
    def num_classes(self):
        return len(self._classes)

    @property
    def classes(self):
        return self._classes

    @property
    def class_colors(self):
        return self._class_colors
