"""Build the C part of the oracle (gcc, host only).  TEST INFRASTRUCTURE."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, 'liboracle_vq.so')


def build(force=False):
    src = os.path.join(HERE, 'vq_argmin.c')
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-fno-fast-math', '-shared', '-fPIC',
                               src, '-o', SO, '-lm'])
    return SO


if __name__ == '__main__':
    print(build(True))
