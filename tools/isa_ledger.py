#!/usr/bin/env python3
"""Instruction ledger of a gfx950 kernel from hipcc's --save-temps assembly: per basic block (label to label) the number of
instructions by class -- MFMA, transcendental VALU, other VALU, SALU, branches, waits, nops, LDS, vector memory -- so that the
inner loops of the issue-bound kernels (csrc/attn.hip) can be held to a budget per tile at compile time, without a GPU.

usage: tools/isa_ledger.py <file.s> <kernel-name-regex> [--blocks] [--range LABEL_FROM LABEL_TO]
  (hipcc --offload-arch=gfx950 -O3 -std=c++17 --save-temps -c mmvid_amd/csrc/attn.hip  ->  attn-hip-amdgcn-amd-amdhsa-gfx950.s)
--blocks prints every basic block; --range sums the blocks from one label up to (not including) another."""
import re
import sys
from collections import OrderedDict

CLASSES = ['mfma', 'trans', 'valu', 'salu', 'branch', 'wait', 'nop', 'lds', 'vmem', 'other']
TRANS = ('v_exp_', 'v_log_', 'v_rcp_', 'v_rsq_', 'v_sqrt_', 'v_sin_', 'v_cos_')


def classify(op):
    if op.startswith('v_mfma') or op.startswith('v_smfmac'):
        return 'mfma'
    if op.startswith(TRANS):
        return 'trans'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('s_cbranch') or op.startswith('s_branch') or op in ('s_endpgm', 's_setpc_b64', 's_swappc_b64'):
        return 'branch'
    if op.startswith('s_waitcnt') or op == 's_barrier':
        return 'wait'
    if op in ('s_nop', 's_sleep'):
        return 'nop'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('buffer_', 'global_', 'scratch_', 'flat_')):
        return 'vmem'
    return 'other'


def kernel_lines(path, pat):
    rx = re.compile(pat)
    out, on, name = [], False, None
    for ln in open(path):
        m = re.match(r'^(_Z\w+|\w+):\s*(;.*)?$', ln)
        if m and not ln.startswith('.L'):
            if on:
                break
            if rx.search(m.group(1)):
                on, name = True, m.group(1)
            continue
        if on:
            if ln.strip().startswith('.section') or ln.strip().startswith('.end_amdhsa_kernel'):
                break
            out.append(ln.rstrip('\n'))
    return name, out


def ledger(lines):
    blocks = OrderedDict()
    cur = 'entry'
    blocks[cur] = dict.fromkeys(CLASSES, 0)
    nops_states = 0
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith(';') or s.startswith('.') and not s.startswith('.LBB'):
            continue
        m = re.match(r'^(\.LBB\w+):', s)
        if m:
            cur = m.group(1)
            blocks[cur] = dict.fromkeys(CLASSES, 0)
            continue
        op = s.split()[0]
        if not re.match(r'^[a-z_0-9]+$', op):
            continue
        blocks[cur][classify(op)] += 1
    return blocks


def fmt(name, c):
    tot = sum(c.values())
    return f'{name:14s} ' + ' '.join(f'{c[k]:6d}' for k in CLASSES) + f' {tot:7d}'


def main():
    path, pat = sys.argv[1], sys.argv[2]
    name, lines = kernel_lines(path, pat)
    if name is None:
        sys.exit(f'no kernel matches {pat}')
    blocks = ledger(lines)
    print(f'kernel {name}')
    print(f'{"block":14s} ' + ' '.join(f'{k:>6s}' for k in CLASSES) + f' {"total":>7s}')
    if '--blocks' in sys.argv:
        for b, c in blocks.items():
            if sum(c.values()):
                print(fmt(b, c))
    if '--range' in sys.argv:
        i = sys.argv.index('--range')
        a, z = sys.argv[i + 1], sys.argv[i + 2]
        tot, on = dict.fromkeys(CLASSES, 0), False
        for b, c in blocks.items():
            if b == a:
                on = True
            if b == z:
                break
            if on:
                for k in CLASSES:
                    tot[k] += c[k]
        print(fmt(f'{a}..{z}', tot))
    tot = dict.fromkeys(CLASSES, 0)
    for c in blocks.values():
        for k in CLASSES:
            tot[k] += c[k]
    print(fmt('whole kernel', tot))


if __name__ == '__main__':
    main()
