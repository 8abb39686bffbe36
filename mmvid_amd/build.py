"""Build libmmvid_hip.so (gfx950) in-tree with hipcc.  `python -m mmvid_amd.build [--force]`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, '_obj')
LIB = os.path.join(HERE, 'libmmvid_hip.so')
SOURCES = ['errors', 'graphs', 'vq', 'gemm', 'norm', 'attn', 'embed', 'optim', 'tower', 'decode', 'decode_persistent', 'conv', 'conv_strip', 'strict', 'sample', 'frontend', 'vqgan', 'probe']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result']


def _deps_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hs.append(os.path.join(os.path.dirname(HERE), 'include', 'mmvid_hip.h'))
    return max(os.path.getmtime(h) for h in hs)


def _compile(name, force, hdr_m):
    src = os.path.join(CSRC, name + '.hip')
    obj = os.path.join(OBJ, name + '.o')
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_m):
        return obj, False
    subprocess.check_call(['hipcc', *FLAGS, '-c', src, '-o', obj])
    return obj, True


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hdr_m = _deps_mtime()
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        res = list(ex.map(lambda n: _compile(n, force, hdr_m), SOURCES))
    objs = [o for o, _ in res]
    if force or any(c for _, c in res) or not os.path.exists(LIB):
        subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', LIB])
        if verbose:
            print('linked', LIB)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
