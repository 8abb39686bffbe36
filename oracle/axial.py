"""Axial positional embedding oracle.  TEST INFRASTRUCTURE.  PARITY UNPINNED:
the arithmetic lives in the third-party package `axial_positional_embedding` (lucidrains,
unpinned in the reference's requirements.txt:2; call sites mmvid_pytorch/dalle_bert.py:326-327,
mmvid_pytorch/modules.py:24-27, mmvid_pytorch/dalle_artv.py:141-146).  Published semantics,
summed mode: parameter weights_i has shape [1, 1.., s_i, ..1, dim]; the table is the
broadcast sum over axes, flattened row-major over axial_shape, truncated to t rows."""
import torch


def axial_table(sd, prefix, axial_shape, dim):
    tot = 0
    n = 1
    for s in axial_shape:
        n *= s
    for i in range(len(axial_shape)):
        w = sd[f'{prefix}.weights_{i}']
        tot = tot + w.expand((1, *axial_shape, dim)).reshape(n, dim)
    return tot  # [prod(shape), dim]


def axial_list_table(sd, prefix, num, axial_shape, dim):
    """mmvid_pytorch/modules.py:8-53 (no [SEP] path): per-frame (h,w) tables concatenated."""
    return torch.cat([axial_table(sd, f'{prefix}.module_list.{v}', axial_shape, dim) for v in range(num)], 0)
