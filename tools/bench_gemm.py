#!/usr/bin/env python3
"""GEMM / conv micro-benchmark on the shapes of the training step (HIP-event timed, TFLOP/s)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import ops

dev = 'cuda'
bf = torch.bfloat16


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def strip():
    """The strip convolution on the encoder's mode-0 layers (54 frames) against the per-tap kernel."""
    for name, N, H, Cin, Cout in (('c128@128', 54, 128, 128, 128), ('c128@64', 54, 64, 128, 128), ('c256@32', 54, 32, 256, 256),
                                  ('c128->256@32', 54, 32, 128, 256)):
        x = torch.randn(N, H, H, Cin, device=dev).to(bf)
        w = (torch.randn(Cout, 9, Cin, device=dev) * 0.03).to(bf)
        b = torch.zeros(Cout, device=dev)
        r32 = torch.randn(N, H, H, Cout, device=dev)
        fl = 2.0 * N * H * H * Cout * 9 * Cin
        t = timeit(lambda: ops.conv2d_nhwc(x, w, b, 0), 5)
        line = f'{name:13s} per-tap {t*1e3:7.1f} us {fl/t/1e9:7.1f} TF |'
        t = timeit(lambda: ops.conv3x3_strip(x, w, b), 5)
        t2 = timeit(lambda: ops.conv3x3_strip(x, w, b, residual=r32, out_dtype=torch.float32), 5)
        line += f' strip: {t*1e3:7.1f} us {fl/t/1e9:7.1f} TF, +res f32: {t2*1e3:7.1f} us {fl/t2/1e9:7.1f} TF |'
        print(line)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'strip':
        return strip()
    M = 10422
    for name, N, K in (('qkv', 2304, 768), ('out', 768, 768), ('fc', 3072, 768), ('proj', 768, 3072)):
        X = torch.randn(M, K, device=dev).to(bf)
        W = (torch.randn(N, K, device=dev) * 0.03).to(bf)
        dY = torch.randn(M, N, device=dev).to(bf)
        dW = torch.zeros(N, K, device=dev)
        fl = 2.0 * M * N * K
        t = timeit(lambda: ops.gemm(X, W))
        print(f'{name:5s} fwd  NT {M}x{N}x{K}: {t*1e3:8.1f} us {fl/t/1e9:8.1f} TF')
        t = timeit(lambda: ops.gemm(dY, W, b_kmajor=True))
        print(f'{name:5s} dX   NN {M}x{K}x{N}: {t*1e3:8.1f} us {fl/t/1e9:8.1f} TF')
        for sk in (None, 3, 4, 8, 14):
            t = timeit(lambda: ops.gemm_dw(dY, X, dW, splitk=sk))
            print(f'{name:5s} dW   TN sk={sk} {N}x{K}x{M}: {t*1e3:8.1f} us {fl/t/1e9:8.1f} TF')
    for name, N, H, Cin, Cout, mode in (('c128@128', 96, 128, 128, 128, 0), ('c256@32', 96, 32, 256, 256, 0), ('c512@8', 96, 8, 512, 512, 0),
                                        ('down128', 96, 128, 128, 128, 1), ('conv_in', 96, 128, 8, 128, 0)):
        x = torch.randn(N, H, H, Cin, device=dev).to(bf)
        w = (torch.randn(Cout, 9, Cin, device=dev) * 0.03).to(bf)
        b = torch.zeros(Cout, device=dev)
        Ho = H // 2 if mode == 1 else H
        fl = 2.0 * N * Ho * Ho * Cout * 9 * Cin
        t = timeit(lambda: ops.conv2d_nhwc(x, w, b, mode), 5)
        print(f'{name:9s} conv N={N} {H}x{H} {Cin}->{Cout}: {t*1e3:8.1f} us {fl/t/1e9:8.1f} TF')


if __name__ == '__main__':
    main()
