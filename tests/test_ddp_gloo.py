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


def _worker(rank, world, port, q):
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
        tr = FlatTrainer(m, order=backward_order, bucket_mb=0.001)  # tiny buckets: several messages per range
        assert tr.world == world
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
        assert tr.sparse_tables and tr._sparse_ranges() == [(0, 512)]  # text_emb [50, 8] = 400 elements, padded to 4 x 128
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
def test_flat_trainer_exchange_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == 'ok' for r in res), res
