"""Data-parallel exchange logic of mmvid_amd.engine on CPU: 2 processes, gloo backend (the GPU path uses the same
code over RCCL).  Checks the flat layout, the backward-ordered early sends, the final sum, and the parameter
broadcast -- without running any kernel (the optimiser update itself is HIP-only)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


class _Blk(nn.Module):
    def __init__(self):
        super().__init__()
        self.w = nn.Parameter(torch.zeros(130, 7))
        self.b = nn.Parameter(torch.zeros(13))


class _Toy(nn.Module):
    """Parameter names shaped like BERT's so backward_order() sees towers layers / heads / tables."""

    def __init__(self):
        super().__init__()
        self.text_emb = nn.Embedding(50, 8)
        self.transformer = nn.Module()
        self.transformer.transformer = nn.Module()
        self.transformer.transformer.resblocks = nn.ModuleList([_Blk() for _ in range(4)])
        self.transformer.on_layers_done = None
        self.to_logits = nn.Sequential(nn.LayerNorm(8), nn.Linear(8, 5))
        self.frozen = nn.Parameter(torch.ones(3), requires_grad=False)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, exchange='allreduce'):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from mmvid_amd.engine import FlatTrainer, backward_order, broadcast_parameters
        torch.manual_seed(100 + rank)
        m = _Toy()
        for p in m.parameters():
            p.data.normal_()
        broadcast_parameters(m)
        ref = [p.detach().clone() for p in m.parameters()]
        tr = FlatTrainer(m, order=backward_order, bucket_mb=0.001, exchange=exchange)  # tiny buckets: several messages per range
        assert tr.world == world and tr.exchange == exchange
        # flat order: tables first, then layers ascending, then heads
        kinds = [0 if not n.startswith(('transformer.', 'to_logits')) else (2 if n.startswith('to_logits') else 1) for n in tr.names]
        assert kinds == sorted(kinds)
        layer_ids = [int(n.split('.')[3]) for n in tr.names if n.startswith('transformer.')]
        assert layer_ids == sorted(layer_ids)
        assert m.transformer.on_layers_done == tr.layers_done
        assert all(o % 128 == 0 for o in tr.offsets)
        # parameters are views into the flat buffer and kept their values
        for p, r in zip(m.parameters(), ref):
            assert torch.equal(p.detach(), r)
        assert m.text_emb.weight.data_ptr() == tr.P.data_ptr()
        for step in range(2):
            tr.zero_grad()
            for i, p in enumerate(tr.params):
                p.grad.fill_(float((rank + 1) * (i + 1) + step))
            # the backward finishes layers 3,2 then 1,0; tables last
            tr.layers_done(2)
            sent_after_first = tr._sent_from
            tr.layers_done(0)
            assert tr._sent_from < sent_after_first < tr.numel
            tr.allreduce_grads()
            assert tr._sent_from == tr.numel and not tr._works
            for i, p in enumerate(tr.params):
                expect = sum((r + 1) * (i + 1) + step for r in range(world))
                assert torch.all(p.grad == expect), (tr.names[i], p.grad.flatten()[0].item(), expect)
        # the update kernel is HIP-only: no silent host fallback
        from mmvid_amd._lib import MMVIDError
        try:
            tr.step()
            q.put((rank, 'step() ran on CPU'))
            return
        except MMVIDError:
            pass
        # every rank ends with identical parameters (broadcast) -> compare a checksum
        s = torch.tensor([float(sum(p.double().sum() for p in m.parameters()))])
        lst = [torch.zeros(1) for _ in range(world)]
        dist.all_gather(lst, s)
        assert all(torch.equal(lst[0], t) for t in lst)
        q.put((rank, 'ok'))
    except Exception as e:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _worker_real(rank, world, port, q):
    """The REAL model (tiny BERT: 2-layer CLIP-shaped tower, all heads, all tables) under FlatTrainer, with the callbacks in
    the order and granularity the real tower backward issues them (OpenAICLIPTransformer.backward_chunks)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from test_host_logic import tiny_bert
        from mmvid_amd.engine import FlatTrainer, backward_order, broadcast_parameters
        torch.manual_seed(7 + rank)
        m = tiny_bert(1, True)
        m.transformer.backward_chunk_layers = 1
        broadcast_parameters(m)
        tr = FlatTrainer(m, order=backward_order, bucket_mb=1, sparse_tables=False)  # the dense path; row-wise exchange: below
        assert m.transformer.on_layers_done == tr.layers_done
        n_train = sum(p.numel() for p in m.parameters() if p.requires_grad)
        assert sum(p.numel() for p in tr.params) == n_train and not any(n.startswith(('vae.', 'cvae.')) for n in tr.names)
        for p, o in zip(tr.params, tr.offsets):
            assert p.data_ptr() == tr.P.data_ptr() + 4 * o and p.grad.data_ptr() == tr.G.data_ptr() + 4 * o
        tr.zero_grad()
        for i, p in enumerate(tr.params):
            p.grad.fill_(float(rank + 1) * (1 + i % 5))
        sent = []
        orig = tr._send

        def spy(lo, hi):
            if hi > lo:
                sent.append((lo, hi))
            return orig(lo, hi)

        tr._send = spy
        chunks = m.transformer.backward_chunks()
        assert chunks == [(1, 2), (0, 1)]
        # the heads' gradients are final before the tower backward starts; each finished chunk extends the sent range down
        for lo, hi in chunks:
            tr.layers_done(lo)
        heads_and_tower = min(o for n, o in zip(tr.names, tr.offsets) if n.startswith(('transformer.', 'to_logits')))
        assert sent[0][1] == tr.numel and sent[-1][0] == heads_and_tower
        assert all(a[0] == b[1] for a, b in zip(sent, sent[1:])), sent  # contiguous, descending, no overlap
        tr.allreduce_grads()
        assert sent[-1][0] == 0 and tr._sent_from == tr.numel
        for i, p in enumerate(tr.params):
            expect = sum((r + 1) * (1 + i % 5) for r in range(world))
            assert torch.all(p.grad == expect), tr.names[i]
        q.put((rank, 'ok'))
    except Exception:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


class _ToySparse(_Toy):
    """_Toy whose text embedding reports the rows a step touched (as BERT.sparse_grad_rows does)."""

    def sparse_grad_rows(self):
        return {'text_emb.weight': self.ids}

    def reset_sparse_grad_rows(self):
        pass


def _worker_sparse(rank, world, port, q):
    """Row-wise exchange of a table gradient: equals the dense all-reduce sum, with ids repeated inside a rank, shared between
    ranks and disjoint; every other parameter still goes through the dense path around the excluded range."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from mmvid_amd.engine import FlatTrainer, backward_order, broadcast_parameters
        torch.manual_seed(5 + rank)
        m = _ToySparse()
        m.ids = None  # nothing touched yet (BERT before its first forward)
        for p in m.parameters():
            p.data.normal_()
        broadcast_parameters(m)
        tr = FlatTrainer(m, order=backward_order, bucket_mb=0.001)
        # nothing logged yet: the table is NOT cut out of the dense all-reduce (it would otherwise never be summed)
        assert tr.sparse_tables and tr._sparse_ranges() == []
        m.ids = torch.tensor([[1]])
        assert tr._sparse_ranges() == [(0, 512)]  # text_emb [50, 8] = 400 elements, padded to 4 x 128
        for step in range(2):
            tr.zero_grad()
            m.ids = torch.tensor([[3, 7, 7, 49], [3, 0, 11 + rank, 20 + 5 * step]])  # 3 shared, 7 repeated, 11 + rank disjoint
            for i, p in enumerate(tr.params):
                p.grad.fill_(float((rank + 1) * (i + 1)))
            g = m.text_emb.weight.grad
            g.zero_()
            for r in m.ids.unique().tolist():  # what a scatter-add of this rank's batch leaves: only its rows are non-zero
                g[r] = torch.arange(8.) + 10 * r + 100 * rank + step
            mine = g.clone()
            tr.layers_done(0)
            tr.allreduce_grads()
            dense = mine.clone()
            dist.all_reduce(dense)  # the reference result: a dense all-reduce of the same table gradient
            assert torch.equal(m.text_emb.weight.grad, dense), (rank, step)
            for i, (n, p) in enumerate(zip(tr.names, tr.params)):
                if n != 'text_emb.weight':
                    assert torch.all(p.grad == sum((r + 1) * (i + 1) for r in range(world))), n
        q.put((rank, 'ok'))
    except Exception:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _worker_dense_fallback(rank, world, port, q):
    """A table whose row log is missing (ids None: no forward logged, or the log overflowed) goes through the dense
    all-reduce: every rank still ends with the full sum."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from mmvid_amd.engine import FlatTrainer, backward_order
        torch.manual_seed(5)
        m = _ToySparse()
        m.ids = None
        tr = FlatTrainer(m, order=backward_order, bucket_mb=0.001)
        tr.zero_grad()
        for i, p in enumerate(tr.params):
            p.grad.fill_(float((rank + 1) * (i + 1)))
        tr.layers_done(0)
        tr.allreduce_grads()
        for i, (n, p) in enumerate(zip(tr.names, tr.params)):
            assert torch.all(p.grad == sum((r + 1) * (i + 1) for r in range(world))), n
        q.put((rank, 'ok'))
    except Exception:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_flat_trainer_dense_fallback_without_row_log_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dense_fallback, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == 'ok' for r in res), res


@pytest.mark.timeout(300)
def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` without a launcher (how the driver runs every N) must start its own ranks
    (train.py:47-66 spawns them from main()); --dry-run = rank / argument plumbing over gloo, no GPU work.  Without
    --dry-run on a box with fewer devices than ranks it must fail AFTER spawning, with a clear message."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--dry-run'],
                       capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['dry_run'] and out['config']['global_batch'] == 12
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                           capture_output=True, text=True, timeout=240, env=env)
        assert r.returncode != 0 and 'device(s) visible' in r.stderr, r.stderr[-2000:]


@pytest.mark.timeout(300)
def test_flat_trainer_sparse_table_exchange_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sparse, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == 'ok' for r in res), res


@pytest.mark.timeout(300)
def test_flat_trainer_real_model_callbacks_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_real, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == 'ok' for r in res), res


@pytest.mark.timeout(300)
def test_flat_trainer_direct_exchange_world4():
    """The direct exchange with four ranks (shards of a quarter; every rank receives three peers' shards and sums them in rank order)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 4, port, q, 'direct')) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == 'ok' for r in res), res


@pytest.mark.timeout(300)
@pytest.mark.parametrize('exchange', ['allreduce', 'direct'])
def test_flat_trainer_exchange_world2(exchange):
    """Both forms of the dense gradient exchange: one all-reduce per message, and the direct form (all_to_all of shards, rank-ordered
    local sum, all-gather: every point-to-point link busy at once) -- the same sums on every rank."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, exchange)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == 'ok' for r in res), res


def test_flat_trainer_loads_torch_adam_state_by_name():
    """a plain torch.optim.Adam state_dict (the reference's checkpoints, train.py:352) is indexed in
    model.parameters() order, the flat buffer in backward order: moments must land on the parameter of the same NAME
    (the tower layers have identical shapes, so an index mix-up would go unnoticed), shapes are verified, and the
    trainer's own state_dict round-trips together with the front-end's (seed, step)."""
    from mmvid_amd.engine import FlatTrainer, backward_order
    torch.manual_seed(0)
    m = _Toy()
    for p in m.parameters():
        p.data.normal_()
    named = [(n, p) for n, p in m.named_parameters() if p.requires_grad]
    opt = torch.optim.Adam([p for _, p in named], lr=3e-4)
    for i, (_, p) in enumerate(named):
        p.grad = torch.full_like(p, float(i + 1))
    opt.step()
    sd = opt.state_dict()
    assert 'names' not in sd
    order = lambda ps: list(reversed(backward_order(ps)))  # any flat order that is not torch's
    tr = FlatTrainer(m, order=order)
    assert tr.names != [n for n, _ in named]  # the two orders really differ
    tr.load_state_dict(sd)
    assert tr.step_count == 1 and tr.lr == 3e-4
    for j, (n, p) in enumerate(named):
        i = tr.names.index(n)
        o, k = tr.offsets[i], p.numel()
        assert torch.equal(tr.M[o:o + k].view(p.shape), sd['state'][j]['exp_avg']), n
        assert torch.equal(tr.V[o:o + k].view(p.shape), sd['state'][j]['exp_avg_sq']), n
    # own format round trip
    own = tr.state_dict()
    tr2 = FlatTrainer(m, order=order)
    tr2.load_state_dict(own)
    assert torch.equal(tr2.M, tr.M) and torch.equal(tr2.V, tr.V)
    # a state saved for other parameters / shapes is refused
    bad = {'state': {0: {'step': torch.tensor(1.), 'exp_avg': torch.zeros(3, 3), 'exp_avg_sq': torch.zeros(3, 3)}},
           'param_groups': sd['param_groups']}
    with pytest.raises(ValueError):
        tr2.load_state_dict(bad)
    with pytest.raises(ValueError):
        tr2.load_state_dict(dict(own, names=own['names'][:-1] + ['nope']))
