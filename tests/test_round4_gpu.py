"""Round 4: the storer-wave GEMM block (csrc/gemm.hip::gemm_bf16_sw_kernel) and what else the round changed, on the GPU."""
import pytest
import torch

from test_models_gpu import DEV, close

pytestmark = pytest.mark.gpu


