"""Round 4 on the GPU: the grouped weight-gradient launch's tile order, the loader-wave GEMM forms, the hardware probes behind the
round's measurements (matrix-pipe ceiling, FETCH_SIZE calibration streams)."""
import ctypes

import pytest
import torch

from test_models_gpu import DEV

pytestmark = pytest.mark.gpu


def test_grouped_weight_gradients_column_major_walk_vs_fp64():
    """The grouped weight-gradient launch walks an output whose X operand is the wider one (c_proj: 768 x 3072) column-major so that
    operand is streamed once, the others row-major.  Every tile runs the whole token reduction: for the tower's four shapes over 12
    layers, with a frozen entry, every output against fp64, and the launch is bit-reproducible."""
    from mmvid_amd import ops
    torch.manual_seed(4)
    G, M = 12, 579
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    kinds, bases = [], []
    for ki, (N, K) in enumerate(shapes):
        dY = (torch.randn(G, M, N, device=DEV) * 0.1).to(torch.bfloat16)
        X = (torch.randn(G, M, K, device=DEV) * 0.1).to(torch.bfloat16)
        base = torch.randn(G, N, K, device=DEV)
        kinds.append((dY, X, [None if (g == 5 and ki == 0) else base[g].clone() for g in range(G)]))
        bases.append(base)
    res = []
    for rep in range(2):
        for (dY, X, outs), base in zip(kinds, bases):
            for g, o in enumerate(outs):
                if o is not None:
                    o.copy_(base[g])
        ops.gemm_dw_multi(kinds, accumulate=True)
        torch.cuda.synchronize()
        res.append([[o.clone() if o is not None else None for o in outs] for _, _, outs in kinds])
    for a, b in zip(res[0], res[1]):
        assert all(x is None and y is None or torch.equal(x, y) for x, y in zip(a, b))
    for (dY, X, outs), base in zip(kinds, bases):
        for g in (0, 5, 11):
            if outs[g] is None:
                continue
            want = torch.einsum('mn,mk->nk', dY[g].double(), X[g].double()) + base[g].double()
            assert ((outs[g].double() - want).abs().max() / want.abs().max()).item() < 2e-5


def test_matrix_pipe_and_stream_probes_run():
    """mmvid_probe 5 (register-only MFMA chains: the sustained matrix-pipe ceiling bench.py reports) and 6 (read streams of known size for
    the FETCH_SIZE calibration) launch, finish and leave the streamed buffer untouched."""
    from mmvid_amd import _lib, ops
    sink = torch.zeros(16, device=DEV)
    for mode in (0, 1):
        _lib.call('mmvid_probe', 5, (ctypes.c_int32 * 3)(50, 256, mode), ops._p(sink), ops._stream())
    buf = torch.arange(1 << 20, device=DEV, dtype=torch.int32)
    ref = buf.clone()
    for mode in (0, 1):
        _lib.call('mmvid_probe', 6, (ctypes.c_int64 * 2)(buf.numel() * 4, mode), ops._p(buf), ops._stream())
    torch.cuda.synchronize()
    assert torch.equal(buf, ref) and float(sink.abs().sum()) == 0.0


def test_attention_outputs_are_batch_order_independent_after_the_swizzle_change():
    """The round-4 LDS swizzle (csrc/attn.hip::swz) changes where a tile's chunks sit in LDS, not what is computed: forward and backward of
    one (sequence, head) do not depend on which other sequences share the launch (bit-identical), with the restricted rows of the BERT mask."""
    from mmvid_amd import _lib, ops
    B, L, H, E = 5, 579, 12, 768
    torch.manual_seed(9)
    qkv = (torch.randn(B * L, 3 * E, device=DEV) * 0.5).bfloat16()
    dO = (torch.randn(B * L, E, device=DEV) * 0.1).bfloat16()

    def run(qkv_, dO_, b):
        out = torch.empty(b * L, E, device=DEV, dtype=torch.bfloat16)
        lse, delta = torch.empty(b * H * L, device=DEV), torch.empty(b * H * L, device=DEV)
        dqkv = torch.empty(b * L, 3 * E, device=DEV, dtype=torch.bfloat16)
        _lib.call('mmvid_attention_fwd', ops._p(qkv_), 3 * E, b, L, H, E, 0.125, 2, 65, 65, 66, 66, ops._p(out), E, ops._p(lse), ops._stream())
        _lib.call('mmvid_attention_bwd', ops._p(qkv_), 3 * E, ops._p(out), E, ops._p(dO_), E, ops._p(lse), ops._p(delta), b, L, H, E, 0.125, 2, 65,
                  65, 66, 66, ops._p(dqkv), 3 * E, ops._stream())
        return out, dqkv

    o_all, g_all = run(qkv, dO, B)
    o_one, g_one = run(qkv[2 * L:3 * L].contiguous(), dO[2 * L:3 * L].contiguous(), 1)
    assert torch.equal(o_all[2 * L:3 * L], o_one) and torch.equal(g_all[2 * L:3 * L], g_one)


def test_sparse_exchange_pack_and_merge_kernels():
    """mmvid_rows_pack / mmvid_rows_merge (what FlatTrainer._exchange_sparse runs on the device instead of torch sort / index_select /
    index_add_): the message of a rank -- ids ascending, repeats blanked, rows zeroed where blanked -- equals the torch formulation bit
    for bit, and merging three peers' messages in rank order equals the dense sum."""
    from mmvid_amd.engine import FlatTrainer
    torch.manual_seed(3)
    V, E, n = 49472, 768, 1152
    W = torch.randn(V, E, device=DEV)
    ids = torch.randint(0, V, (n, ), device=DEV)
    ids[100:400] = ids[:300].clone()  # repeats
    uid, rows = FlatTrainer.pack_rows(W, ids)
    srt, _ = torch.sort(ids)
    first = torch.ones_like(srt, dtype=torch.bool)
    first[1:] = srt[1:] != srt[:-1]
    assert torch.equal(uid, torch.where(first, srt, torch.full_like(srt, -1)))
    assert torch.equal(rows, W.index_select(0, srt) * first.unsqueeze(1).to(W.dtype))
    # merge: this rank is 1 of 4; peers 0, 2, 3 bring their own messages (overlapping ids)
    peers = []
    for r in range(4):
        pid = torch.randint(0, 2000, (n, ), device=DEV)  # dense in a small range: many rows get several contributions
        Wp = torch.zeros(V, E, device=DEV)
        Wp[pid.unique()] = torch.randn(pid.unique().numel(), E, device=DEV)
        peers.append((Wp, ) + FlatTrainer.pack_rows(Wp, pid))
    mine = peers[1][0].clone()
    all_ids = torch.cat([p[1] for p in peers])
    all_rows = torch.cat([p[2] for p in peers])
    FlatTrainer.merge_rows(mine, all_ids, all_rows, 1, n)
    want = peers[1][0] + peers[0][0] + peers[2][0] + peers[3][0]  # rank order 0, 2, 3 added to rank 1's own
    assert torch.equal(mine, want)


@pytest.mark.parametrize('n,h,c,dt,swish,out', [(54, 8, 512, torch.float32, True, torch.bfloat16), (7, 8, 512, torch.bfloat16, True, torch.bfloat16),
                                                (3, 16, 256, torch.float32, False, torch.float32), (5, 4, 128, torch.float32, True, torch.bfloat16),
                                                (2, 16, 32, torch.bfloat16, True, torch.bfloat16), (2, 8, 64, torch.bfloat16, True, torch.bfloat16)])
def test_small_map_groupnorm_single_launch_vs_torch(n, h, c, dt, swish, out):
    """GroupNorm of a map of <= 256 pixels without fused statistics (the VQGAN's 8x8 / 16x16 levels): one launch, one block per image
    (statistics, finalisation, apply) -- against torch, and bit-reproducible."""
    import torch.nn.functional as F
    from mmvid_amd import ops
    torch.manual_seed(n + h + c)
    x = (torch.randn(n, h, h, c, device=DEV) * 2 + 0.5).to(dt)
    w, b = torch.randn(c, device=DEV), torch.randn(c, device=DEV)
    res = {1: ops.groupnorm_swish(x, w, b, swish=swish, out_dtype=out), 0: ops.groupnorm_swish(x, w, b, swish=swish, out_dtype=out)}
    assert torch.equal(res[1], res[0])
    ref = F.group_norm(x.float().permute(0, 3, 1, 2), 32, w, b, 1e-6)
    if swish:
        ref = ref * torch.sigmoid(ref)
    ref = ref.permute(0, 2, 3, 1)
    err = ((res[1].float() - ref).abs().max() / ref.abs().max()).item()
    assert err < (1e-2 if out == torch.bfloat16 else 1e-5), err


@pytest.mark.parametrize('B,P', [(1, 2), (1, 250), (2, 250), (2, 7)])
def test_persistent_decode_step_matches_the_five_launch_form(B, P):
    """mmvid_tower_decode_persistent (csrc/decode_persistent.hip): one launch of 256 co-resident blocks per token, values handed between
    blocks as tagged words.  Same rounding points as the five-launches-per-layer step: hidden states and the key/value cache agree to
    fp32-summation-order / one-bf16-ulp level over 24 positions (eager, then captured and replayed), attention ranges of 0 (prompt of 2 positions: an empty
    range at batch 1) .. 68 keys per range, every supported batch size; no poll ever timed out (workspace word 1)."""
    from mmvid_amd import _lib
    from mmvid_amd.clip_tower import OpenAICLIPTransformer
    from test_models_gpu import close
    torch.manual_seed(B)
    L, steps = 320, 24
    tw = OpenAICLIPTransformer(seq_len=L, which_model='openai_clip_visual', causal=True, layers=3).to(DEV).eval()
    x = torch.randn(B, P + steps, 768, device=DEV) * 0.5
    with torch.no_grad():
        caches = [tw.new_kv_cache(B, L, DEV) for _ in range(2)]
        for c in caches:
            tw.prefill(x[:, :P].contiguous(), c)
        ref = tw.decode_session(caches[0], P, fused='launches')
        per = tw.decode_session(caches[1], P, graph=True)
        assert per.persistent and not ref.persistent
        for k in range(steps):
            hr = ref.step(x[:, P + k].contiguous()).clone()
            hp = per.step(x[:, P + k].contiguous()).clone()
            close(hp, hr, 1e-2, f'B={B}: persistent vs five-launch decode step at position {P + k}')
        assert per.graph is not None
        close(caches[1][:, :, :P + steps], caches[0][:, :, :P + steps], 1e-2, 'key/value cache')
        assert int(per.ws[1]) == 0 and int(per.ws[0]) == steps
        per.check()
        # a raised failure flag (a poll that timed out): the kernel returns at once from then on; step() notices, repeats the position
        # with the five-launch form and stays there (round 5: it used to hand back a void hidden state until somebody called check())
        extra = torch.randn(B, 768, device=DEV) * 0.5
        hr = ref.step(extra).clone()
        per.ws[1] = 1
        assert per.host_pos == P + steps
        hp = per.step(extra, verify=True).clone()
        assert not per.persistent and per.fell_back == 1
        close(hp, hr, 1e-2, 'the step after a failed persistent launch (five-launch fall-back)')
        per.check()
        # without verification the caller keeps its own restart point: the failure stays visible
        per2 = tw.decode_session(caches[1], P + steps + 1, graph=False)
        per2.ws[1] = 1
        per2.step(extra)  # (the default: no host read)
        assert per2.failed()
        with pytest.raises(_lib.MMVIDError, match='timed out'):
            per2.check()


@pytest.mark.parametrize('R,V', [(1, 1024), (16, 1024), (3, 1000), (5, 4099)])
def test_token_draw_one_block_per_row_equals_the_wave_per_row_kernel(R, V):
    """mmvid_sample_race_at without y (the ART-V draw): one block per row, a thread's elements requested together.  Same keys E / expf(x - max),
    same (key, index) order: the tokens are the wave-per-row kernel's (the form the oracle parity tests pin) bit for bit, with near-ties and
    -inf logits in the rows; with a device-side draw index the variates are block (index - step0) of a pre-drawn [draws, R, V] tensor."""
    from mmvid_amd import ops
    torch.manual_seed(R * 7 + V)
    logits = torch.randn(R, V, device=DEV) * 3
    logits[:, ::7] = float('-inf')
    logits[:, 5] = logits[:, 11]  # equal probabilities: the lower E wins, and equal keys the lower index
    E3 = torch.empty(6, R, V, device=DEV).exponential_()
    E3[:, :, 5] = E3[:, :, 11]
    for d in (0, 4):
        ref, _ = ops.sample_race(logits, E3[d].contiguous(), None, 0.0, logit_div=0.7, want_y=True)   # wave per row (with y)
        tok, y = ops.sample_race(logits, E3[d].contiguous(), None, 0.0, logit_div=0.7, want_y=False)  # block per row
        assert y is None and torch.equal(tok, ref)
        pos = torch.tensor([100 + d], dtype=torch.int32, device=DEV)
        out = torch.full((R, ), -1, dtype=torch.long, device=DEV)
        ops.sample_race(logits, E3, None, 0.0, logit_div=0.7, want_y=False, tok_out=out, step_dev=pos, step0=100)
        assert torch.equal(out, ref)


@pytest.mark.parametrize('B', [1, 2])
def test_artv_token_step_in_one_launch(B):
    """mmvid_artv_token_step_persistent: [embedding row of the token drawn last -> persistent tower step -> LN + head block -> draw of the
    next token -> position + 1] as ONE launch.  Against the same chain as separate launches from the same state: the embedded token is
    filed in the record, the hidden state and the head's logits agree to fp32-summation-order / bf16-ulp level, the token drawn IS the
    draw kernel's on the logits the launch produced (same rule, bit for bit), the position has advanced; three consecutive steps."""
    from mmvid_amd import _lib, ops
    from mmvid_amd.clip_tower import OpenAICLIPTransformer
    from test_models_gpu import close
    torch.manual_seed(40 + B)
    L, P, V, E_, steps = 320, 200, 1024, 768, 3
    tw = OpenAICLIPTransformer(seq_len=L, which_model='openai_clip_visual', causal=True, layers=3).to(DEV).eval()
    table = torch.randn(1500, E_, device=DEV) * 0.5
    pos_rows = torch.randn(L + 4, E_, device=DEV) * 0.1
    lnw, lnb = torch.rand(E_, device=DEV) + 0.5, torch.randn(E_, device=DEV) * 0.1
    w_blk = (torch.randn(V, E_, device=DEV) * 0.05).bfloat16()
    b_blk = torch.randn(V, device=DEV) * 0.1
    Eall = torch.empty(8, B, V, device=DEV).exponential_()
    tok0 = torch.randint(0, 1500, (B, ), device=DEV)
    x = torch.randn(B, P, E_, device=DEV) * 0.5
    with torch.no_grad():
        caches = [tw.new_kv_cache(B, L, DEV) for _ in range(2)]
        for c in caches:
            tw.prefill(x, c)
        ref = tw.decode_session(caches[0], P, graph=False, fused='launches')
        one = tw.decode_session(caches[1], P, graph=False)
        assert one.persistent
        tok_r, tok_o = tok0.clone(), tok0.clone()
        record = torch.full((B, 8), -1, dtype=torch.long, device=DEV)
        logits_o = torch.empty(B, V, device=DEV)
        tk = _lib.DecodeToken()
        tk.tok, tk.table, tk.table_rows, tk.pos_rows, tk.pos_off = tok_o.data_ptr(), table.data_ptr(), 1500, pos_rows.data_ptr(), 2
        tk.record, tk.record_ld, tk.record_pos0 = record.data_ptr(), 8, P
        tk.lnf_w, tk.lnf_b, tk.lnf_eps, tk.head_w, tk.head_b, tk.V = lnw.data_ptr(), lnb.data_ptr(), 1e-5, w_blk.data_ptr(), b_blk.data_ptr(), V
        tk.E, tk.e_step_stride, tk.e_pos0, tk.temperature, tk.tok_offset, tk.logits_out = Eall.data_ptr(), B * V, P - 1, 0.9, 0, logits_o.data_ptr()
        for n in range(steps):
            emb = torch.empty(B, E_, device=DEV)
            ops.decode_embed(tok_r, table, pos_rows, ref.pos, emb, pos_off=2)
            before = tok_o.clone()
            h_r = ref.step(emb).clone()
            logits_r = torch.empty(B, V, device=DEV)
            ops.gemv_rows(h_r, w_blk, b_blk, ln=(lnw, lnb, 1e-5), round_in=True, out=logits_r)
            one.token_step(tk)
            torch.cuda.synchronize()
            close(one.y, h_r, 1e-2, f'B={B} step {n}: hidden state')
            close(logits_o, logits_r, 1e-2, f'B={B} step {n}: head logits')
            assert torch.equal(record[:, n], before) and int(one.pos) == P + n + 1
            want, _ = ops.sample_race(logits_o, Eall[n + 2].contiguous(), None, 0.0, logit_div=0.9, want_y=False)
            assert torch.equal(tok_o, want), (n, tok_o, want)
            tok_r.copy_(tok_o)  # keep the two chains on the same trajectory
        one.check()


@pytest.mark.timeout(600)
def test_two_ranks_sharing_this_gpu_run_the_world2_step():
    """The world > 1 GPU path, end to end through bench.py on a 1-GPU box (MMVID_BENCH_SHARED_GPU=1: both ranks on the devices that
    exist, exchange through gloo because RCCL refuses two ranks on one device; never a measurement): per-rank seeds, weight broadcast, the
    backward split into chunks with the bucketed exchange between them, the row-wise table exchange with its pack / merge kernels, clip +
    Adam on the averaged gradients, max-over-ranks timing, ONE JSON line.  gloo cannot be captured: `collectives_capturable` says so BEFORE
    the step is captured (a collective failing inside a capture would leave the group's streams in capture mode and take the eager
    fall-back down with it), and the step runs eagerly."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['MMVID_BENCH_SHARED_GPU'] = '1'
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29517', os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--no-cpu-baseline', '--layers', '2'], capture_output=True, text=True, timeout=540, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 12 and 'SHARE' in out['config']['parallelism']
    assert out['config']['step_launch'].startswith('eager (the gloo backend cannot capture')
    assert out['loss'] == out['loss'] and out['gradient_exchange']['row_wise_tables'] == ['text_emb.weight']
