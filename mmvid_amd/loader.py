"""`from mmvid_pytorch.loader import TextVideoDataset` (utils_train.py:25) -> the same import path here.  The other dataset
classes of the reference's loader.py / loader_ext.py (mp4, image stacks, shape-attribute, VoxCeleb, iPER) are not built."""
from .data import TextVideoDataset  # noqa: F401
