"""`from mmvid_pytorch.loader import TextVideoDataset` / `TextMP4Dataset` / `TextImageStackDataset` (utils_train.py:25, 46, 64) -> the
same import path here.  (`TextImageDataset`, the shape-attribute and iPER loaders of the reference are not built.)"""
from .data import TextVideoDataset  # noqa: F401
from .loader_files import TextImageStackDataset, TextMP4Dataset  # noqa: F401
