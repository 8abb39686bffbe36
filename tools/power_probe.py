#!/usr/bin/env python3
"""Is the MFMA work of the step POWER-bound?  Samples the GPU's hwmon power / cap and the shader clock (sysfs; rocm-smi as a fallback)
while (a) one GEMM shape runs back to back for ~2 s, (b) the same GEMM with ZERO operands (no toggling in the matrix pipe), (c) an
HBM-bound copy.  Prints mean power, the cap, mean sclk and the achieved TFLOP/s of each phase."""
import glob
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import ops

dev, bf = 'cuda', torch.bfloat16


def find(pattern):
    g = glob.glob(pattern)
    return g[0] if g else None


PWS = sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_input'))
SCLKS = [os.path.join(os.path.dirname(p), 'freq1_input') for p in PWS]
CAP = find('/sys/class/drm/card*/device/hwmon/hwmon*/power1_cap')
PW, SCLK = (PWS[0], SCLKS[0]) if PWS else (None, None)
print('hwmon nodes:', len(PWS))
if CAP:
    print('power cap (W):', int(open(CAP).read()) / 1e6)


def rd(path, scale):
    try:
        return int(open(path).read()) / scale
    except Exception:
        return float('nan')


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop, self.p, self.f = False, [], []

    def run(self):  # the box exposes every GPU of the node; ours is the one whose power moves: keep the per-sample maximum
        while not self.stop:
            if PWS:
                vals = [(rd(p, 1e6), rd(f, 1e6)) for p, f in zip(PWS, SCLKS)]
                best = max(vals, key=lambda v: v[0] if v[0] == v[0] else -1)
                self.p.append(best[0]), self.f.append(best[1])
            time.sleep(0.01)


def phase(name, fn, flops, seconds=2.0):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s = Sampler()
    s.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    dt = time.time() - t0
    s.stop = True
    s.join()
    import statistics as st
    p = st.mean(s.p[len(s.p) // 4:]) if s.p else float('nan')
    f = st.mean(s.f[len(s.f) // 4:]) if s.f else float('nan')
    print(f'{name:44s} {dt / n * 1e6:8.1f} us/call  {flops * n / dt / 1e12:8.1f} TFLOP/s   power {p:7.1f} W   sclk {f:7.0f} MHz')


M, N, K = 10422, 3072, 768
X = torch.randn(M, K, device=dev).to(bf)
W = (torch.randn(N, K, device=dev) * 0.03).to(bf)
out = torch.empty(M, N, device=dev, dtype=bf)
X0, W0 = torch.zeros_like(X), torch.zeros_like(W)
fl = 2.0 * M * N * K
if not PW:
    print(subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True).stdout[-1500:])
import ctypes
from mmvid_amd import _lib
sink = torch.zeros(16, device=dev)
for mode, label in ((1, 'random bits'), (0, 'zeros')):
    for blocks in (256, 512):
        arr = (ctypes.c_int32 * 3)(2000, blocks, mode)
        fl_p = 2.0 * 32 * 32 * 16 * 32 * 2000 * 8 * blocks
        phase(f'register-only MFMA chains, {blocks} blocks x 8 waves, {label}', lambda arr=arr: _lib.call('mmvid_probe', 5, arr, ops._p(sink), ops._stream()), fl_p, 1.5)
# the same chains with R operand-fragment reads from LDS per 4 MFMAs (mode bits 4-7): 4 = what a 2 x 2 register tile per wave reads (1 KiB
# per MFMA), 3 = a 4 x 2 tile, 2 = a 4 x 4 tile -- how much of the power budget the operand reads take
for R in (4, 3, 2):
    arr = (ctypes.c_int32 * 3)(2000, 256, 1 | (R << 4))
    fl_p = 2.0 * 32 * 32 * 16 * 32 * 2000 * 8 * 256
    phase(f'MFMA chains + {R} LDS fragment reads per 4 MFMAs, random bits', lambda arr=arr: _lib.call('mmvid_probe', 5, arr, ops._p(sink), ops._stream()), fl_p, 1.5)
for R in (0, 3):
    arr = (ctypes.c_int32 * 3)(2000, 256, 1 | (R << 4) | 0x100)
    fl_p = 2.0 * 32 * 32 * 16 * 32 * 2000 * 4 * 256
    phase(f'ONE wave per SIMD: MFMA chains + {R} LDS fragment reads per 4 MFMAs, random bits', lambda arr=arr: _lib.call('mmvid_probe', 5, arr, ops._p(sink), ops._stream()), fl_p, 1.5)
if os.environ.get('PROBE_ONLY'):
    sys.exit(0)
phase('idle (torch.cuda.synchronize only)', lambda: None, 0, 0.5)
phase('GEMM 10422x3072x768 random operands', lambda: ops.gemm(X, W, out=out), fl)
phase('GEMM 10422x3072x768 zero operands', lambda: ops.gemm(X0, W0, out=out), fl)
big = torch.empty(256 << 20, device=dev, dtype=torch.uint8)
big2 = torch.empty_like(big)
phase('copy 256 MiB (HBM-bound)', lambda: big2.copy_(big), 0)
K2 = 3072
X2 = torch.randn(M, K2, device=dev).to(bf)
W2 = (torch.randn(768, K2, device=dev) * 0.03).to(bf)
o2 = torch.empty(M, 768, device=dev, dtype=bf)
phase('GEMM 10422x768x3072 random operands', lambda: ops.gemm(X2, W2, out=o2), 2.0 * M * 768 * K2)
X20, W20 = torch.zeros_like(X2), torch.zeros_like(W2)
phase('GEMM 10422x768x3072 zero operands', lambda: ops.gemm(X20, W20, out=o2), 2.0 * M * 768 * K2)
