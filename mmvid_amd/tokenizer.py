"""`from mmvid_pytorch.tokenizer import SimpleTokenizer` (utils_train.py:187, utils_eval.py:241) -> the same import path here."""
from .data import SimpleTokenizer  # noqa: F401
