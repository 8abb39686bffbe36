"""Round 4: the storer-wave GEMM block (csrc/gemm.hip::gemm_bf16_sw_kernel) and what else the round changed, on the GPU."""
import pytest
import torch

from test_models_gpu import DEV, close

pytestmark = pytest.mark.gpu


def _gelu(x):
    return x * torch.sigmoid(1.702 * x)


def _gelu_grad(x):
    s = torch.sigmoid(1.702 * x)
    return s * (1 + 1.702 * x * (1 - s))


@pytest.mark.parametrize('M,N,K', [(10422, 2304, 768), (10422, 3072, 768), (3000, 3072, 64), (2900, 3072, 128), (4100, 2304, 200),
                                   (70000, 128, 96)])
@pytest.mark.parametrize('kind', ['plain', 'bias', 'gelu', 'dx', 'dact'])
def test_storer_wave_gemm_matches_torch_and_the_round3_kernel(M, N, K, kind):
    """Persistent bf16-output GEMMs hand every finished tile to storer waves through LDS (option gemm_sw).  Against fp32 torch on
    the same bf16 operands, against the round-3 kernel (gemm_sw 0: same products, so at most the last bf16 bit of an output may
    differ where the epilogue's rounding order changed), bit-reproducible launch after launch, ragged M, short K (1-4 K tiles:
    the storers' schedule has few barriers to spread its stores over), with and without bias."""
    from mmvid_amd import _lib, ops
    torch.manual_seed(M % 97 + N + K)
    bf = torch.bfloat16
    X = torch.randn(M, K, device=DEV).to(bf)
    bias = torch.randn(N, device=DEV) * 0.5 if kind in ('bias', 'gelu') else None
    if kind in ('dx', 'dact'):  # dX = dY W with W k-major [K(red)][N(out)]
        W = (torch.randn(K, N, device=DEV) * 0.05).to(bf)
        ref = X.float() @ W.float()
    else:
        W = (torch.randn(N, K, device=DEV) * 0.05).to(bf)
        ref = X.float() @ W.float().t()
    if bias is not None:
        ref = ref + bias
    pre_saved = (torch.randn(M, N, device=DEV) * 1.5).to(bf) if kind == 'dact' else None

    def run():
        if kind == 'gelu':
            pre = torch.empty(M, N, device=DEV, dtype=bf)
            y = ops.gemm(X, W, bias=bias, act=1, save_pre=pre)
            return y, pre, None
        if kind == 'dact':
            cs = torch.zeros(N, device=DEV)
            y = ops.gemm(X, W, b_kmajor=True, dact_pre=pre_saved, colsum=cs)
            return y, None, cs
        if kind == 'dx':
            return ops.gemm(X, W, b_kmajor=True), None, None
        return ops.gemm(X, W, bias=bias), None, None

    outs = {}
    try:
        for sw in (1, 0):
            _lib.call('mmvid_set_option', b'gemm_sw', sw)
            a = run()
            b = run()
            torch.cuda.synchronize()
            assert torch.equal(a[0], b[0]), 'not reproducible'
            outs[sw] = a
    finally:
        _lib.call('mmvid_set_option', b'gemm_sw', 1)
    y, pre, cs = outs[1]
    y0, pre0, cs0 = outs[0]
    scale = ref.abs().max().item()
    if kind == 'gelu':
        close(pre, ref, 1e-2, 'pre')
        assert torch.equal(y, _gelu(pre.float()).to(bf)) or (y.float() - _gelu(pre.float())).abs().max().item() <= 2e-2 * scale
        close(y, _gelu(ref), 1.5e-2, 'gelu out')
        close(y, y0, 1.2e-2, 'vs round-3 kernel')
    elif kind == 'dact':
        want = ref * _gelu_grad(pre_saved.float())
        close(y, want, 1.5e-2, 'dact out')
        close(y, y0, 1.2e-2, 'vs round-3 kernel')
        close(cs, want.sum(0), 1e-2, 'column sums')
        close(cs, cs0, 1e-2, 'column sums vs round-3 kernel')
    else:
        close(y, ref, 1e-2, 'out')
        if bias is None:
            assert torch.equal(y, y0), f'{(y != y0).sum().item()} of {y.numel()} outputs differ from the round-3 kernel'
        else:
            close(y, y0, 1e-2, 'vs round-3 kernel')
            assert (y != y0).float().mean().item() < 0.05  # the bias enters the fp32 sum first instead of last: rare last-bit flips
