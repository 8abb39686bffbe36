#!/usr/bin/env python3
"""The eight GEMM calls of one tower layer exactly as tower.hip issues them at the training step's size (M = 18 x 579 rows) -- bias,
QuickGELU + saved pre-activation, fp32 residual in / out, QuickGELU' operand + fused column sums, bf16 dX outputs -- HIP-event timed,
TFLOP/s.  (tools/bench_gemm.py times bare X W^T products; the round-4 review asked for like-with-like against the in-situ numbers.)

    python tools/bench_gemm_step.py [option value ...]     e.g.  tools/bench_gemm_step.py gemm_groupn 0 1
With an option: one column per value of that library option (mmvid_set_option)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import _lib, ops
from bench_gemm import timeit

dev, bf, f32 = 'cuda', torch.bfloat16, torch.float32
M, E, F = 18 * 579, 768, 3072
torch.manual_seed(0)


def rb(*s, scale=0.5):
    return (torch.randn(*s, device=dev) * scale).to(bf)


h, o, act, g16, dpre, dqkv = rb(M, E), rb(M, E), rb(M, F), rb(M, E, scale=0.1), rb(M, F, scale=0.1), rb(M, 3 * E, scale=0.1)
pre = rb(M, F)
x32 = torch.randn(M, E, device=dev)
w_in, w_out, w_fc, w_pj = rb(3 * E, E, scale=0.03), rb(E, E, scale=0.03), rb(F, E, scale=0.03), rb(E, F, scale=0.03)
b_in, b_out, b_fc, b_pj = (torch.randn(n, device=dev) * 0.1 for n in (3 * E, E, F, E))
o_qkv, o_x, o_pre, o_act = torch.empty(M, 3 * E, device=dev, dtype=bf), torch.empty(M, E, device=dev), torch.empty(M, F, device=dev, dtype=bf), torch.empty(M, F, device=dev, dtype=bf)
o_dpre, o_dh, o_do = torch.empty(M, F, device=dev, dtype=bf), torch.empty(M, E, device=dev, dtype=bf), torch.empty(M, E, device=dev, dtype=bf)
cs = torch.zeros(F, device=dev)

CALLS = [
    ('fwd qkv      (bias, bf16 out)', 3 * E, E, lambda: ops.gemm(h, w_in, bias=b_in, out=o_qkv)),
    ('fwd out-proj (bias, +x fp32, fp32 out)', E, E, lambda: ops.gemm(o, w_out, bias=b_out, residual=x32, out=o_x)),
    ('fwd c_fc     (bias, QuickGELU, pre saved)', F, E, lambda: ops.gemm(h, w_fc, bias=b_fc, act=1, save_pre=o_pre, out=o_act)),
    ('fwd c_proj   (bias, +x fp32, fp32 out)', E, F, lambda: ops.gemm(act, w_pj, bias=b_pj, residual=x32, out=o_x)),
    ("dX  d_pre    (x QuickGELU'(pre), colsum)", F, E, lambda: ops.gemm(g16, w_pj, b_kmajor=True, dact_pre=pre, out=o_dpre, colsum=cs)),
    ('dX  c_fc     (bf16 out)', E, F, lambda: ops.gemm(dpre, w_fc, b_kmajor=True, out=o_dh)),
    ('dX  out-proj (bf16 out)', E, E, lambda: ops.gemm(g16, w_out, b_kmajor=True, out=o_do)),
    ('dX  in-proj  (bf16 out)', E, 3 * E, lambda: ops.gemm(dqkv, w_in, b_kmajor=True, out=o_dh)),
]


def main():
    opt = sys.argv[1] if len(sys.argv) > 2 else None
    vals = [int(v) for v in sys.argv[2:]] if opt else [0]
    print(f'{"call":44s} ' + ' '.join(f'{(opt or "default") + " " + str(v):>18s}' for v in vals))
    tot = [0.0] * len(vals)
    for name, N, K, fn in CALLS:
        fl = 2.0 * M * N * K
        row = f'{name:44s} '
        for i, v in enumerate(vals):
            if opt:
                _lib.call('mmvid_set_option', opt.encode(), v)
            t = timeit(fn, 30)
            tot[i] += t
            row += f'{t * 1e3:8.1f} us {fl / t / 1e9:6.0f} TF '
        print(row)
    fl = 2.0 * M * (3 * E * E + E * E + 2 * E * F) * 2
    print(f'{"one layer, forward + dX":44s} ' + ' '.join(f'{t * 1e3:8.1f} us {fl / t / 1e9:6.0f} TF ' for t in tot))


if __name__ == '__main__':
    main()
