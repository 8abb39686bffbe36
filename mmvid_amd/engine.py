"""Training engine for the data-parallel hot path (train.py:28-35, 298-325 of the reference: DDP + clip_grad_norm_
+ Adam), MI355X-first:

  * every trainable parameter, its gradient and its Adam moments are slices of FLAT fp32 buffers (plus one flat
    bf16 shadow the MFMA kernels read), so the optimiser is one HIP launch over 125 M elements and the gradient
    exchange is a handful of large RCCL collectives instead of hundreds of per-tensor ones;
  * gradients are all-reduced (SUM; the 1/world factor is folded into the Adam kernel) in buckets on a side
    stream.  Buckets follow the order in which the hand-written backward FINISHES gradients -- heads and tower
    layers first, embedding tables last -- so the exchange of the big early buckets overlaps the rest of the
    backward.  xGMI is point-to-point: few, large messages keep every link busy.

One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests.
"""
import os

import torch
import torch.distributed as dist

from . import _lib, ops

ALIGN = 128  # elements: keeps every slice 16-B aligned for the vector kernels (fp32 and bf16)


def _round_up(x, m):
    return (x + m - 1) // m * m


class WarmupLR:
    """The reference's default schedule (utils_train.py:373-385 -> deepspeed.runtime.lr_schedules.WarmupLR; third-party,
    restated from its published semantics: parity unpinned): lr = min + (max - min) * gamma, gamma = log(k + 1) /
    log(warmup) for scheduler step k < warmup, else 1.  train.py:373-374 steps it after every `every`-th iteration;
    before its first step the optimiser runs at its construction lr.  `lr_at(i)` is the host view; the training step
    evaluates the same closed form on the device (mmvid_lr_schedule) so that a captured step follows it."""

    def __init__(self, warmup_min_lr=1e-6, warmup_max_lr=1e-4, warmup_num_steps=5000, every=1):
        self.min_lr, self.max_lr = float(warmup_min_lr), float(warmup_max_lr)
        self.warmup = max(2, int(warmup_num_steps))
        self.every = max(1, int(every))

    def lr_at(self, iteration):
        import math
        ns = iteration // self.every
        if ns == 0:
            return self.max_lr
        k = ns - 1
        gamma = math.log(k + 1) / math.log(self.warmup) if k < self.warmup else 1.0
        return self.min_lr + (self.max_lr - self.min_lr) * gamma


class FlatTrainer:
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0,
                 process_group=None, bucket_mb=64, order=None, lr_schedule=None, force_exchange=False, sparse_tables=True,
                 lazy_rows=True, exchange=None):
        self.model = model
        # how a dense gradient message is summed over the ranks: 'allreduce' (one RCCL all-reduce: a ring, bound by ONE xGMI link
        # per hop) or 'direct' (every rank sends shard j of the message straight to rank j -- all_to_all over the 7
        # point-to-point links at once --, sums the shards it received in rank order, and the summed shards are all-gathered).
        # Same sums up to fp32 summation order; MMVID_EXCHANGE overrides.
        self.exchange = os.environ.get('MMVID_EXCHANGE', exchange or 'allreduce')
        assert self.exchange in ('allreduce', 'direct'), self.exchange
        self._direct_buf = None
        # row-wise exchange of the tables model.sparse_grad_rows() names (MMVID_SPARSE_TABLES=0 forces the dense all-reduce)
        self.sparse_tables = bool(sparse_tables) and os.environ.get('MMVID_SPARSE_TABLES', '1') != '0'
        self.lr_schedule = lr_schedule
        # force_exchange: run the all-reduce path even in a group of one (exercises RCCL + graph capture on a 1-GPU box)
        self.force_exchange = bool(force_exchange) and dist.is_available() and dist.is_initialized()
        self.exchange_enabled = True
        params = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if order is not None:
            params = order(params)
        self.names = [n for n, _ in params]
        self.params = [p for _, p in params]
        dev = self.params[0].device
        offs, tot = [], 0
        for p in self.params:
            offs.append(tot)
            tot += _round_up(p.numel(), ALIGN)
        self.offsets, self.numel = offs, tot
        self.P = torch.zeros(tot, device=dev, dtype=torch.float32)
        self.G = torch.zeros(tot, device=dev, dtype=torch.float32)
        self.M = torch.zeros(tot, device=dev, dtype=torch.float32)
        self.V = torch.zeros(tot, device=dev, dtype=torch.float32)
        self.S = torch.zeros(tot, device=dev, dtype=torch.bfloat16) if dev.type == 'cuda' else None
        for p, o in zip(self.params, offs):
            n = p.numel()
            self.P[o:o + n].copy_(p.detach().reshape(-1))
            p.data = self.P[o:o + n].view(p.shape)
            p.grad = self.G[o:o + n].view(p.shape)
        if self.S is not None:
            ops.cast_bf16(self.P, self.S)
            self._attach_shadows()
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.step_count = 0
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.bucket_elems = max(ALIGN, int(bucket_mb * (1 << 20) / 4) // ALIGN * ALIGN)
        self._sq = torch.zeros(1, device=dev, dtype=torch.float32)
        self._sq_partials = torch.zeros(2048, device=dev, dtype=torch.float32) if dev.type == 'cuda' else None
        # the step count also lives on the device, advanced by a device op: a captured step (GraphedStep) replays with
        # the right Adam bias corrections and learning rate without the host passing new scalars
        self._step_dev = torch.zeros(1, device=dev, dtype=torch.float32) if dev.type == 'cuda' else None
        self._lr_dev = torch.full((1, ), float(lr), device=dev, dtype=torch.float32) if dev.type == 'cuda' else None
        self._comm_stream = torch.cuda.Stream(device=dev) if dev.type == 'cuda' else None
        self._works = []
        self._sent_from = tot  # gradients at flat offsets >= this are already on the wire this step
        # first flat offset of each tower layer (the flat order puts layers ascending, heads after, tables before)
        self._layer_start = {}
        for n, o in zip(self.names, offs):
            if n.startswith('transformer.transformer.resblocks.'):
                i = int(n.split('.')[3])
                self._layer_start[i] = min(o, self._layer_start.get(i, tot))
        tw = getattr(model, 'transformer', None)
        # (only when there is something to exchange: a chunked backward returns to the host every few layers, and a group of one
        # would pay for that -- the weight gradients of all layers could no longer go out as one launch per kind, csrc/tower.hip)
        if tw is not None and hasattr(tw, 'on_layers_done') and self._layer_start and (self.world > 1 or self.force_exchange):
            tw.on_layers_done = self.layers_done
        self._init_lazy_rows(lazy_rows and os.environ.get('MMVID_LAZY_ROWS', '1') != '0')

    # ------------------------------------------------------------------------------------------ lazy table rows
    def _init_lazy_rows(self, enabled):
        """The table `model.sparse_grad_rows()` names (BERT: text_emb, 49,472 x 768 = 30 % of the parameters) gets a flag per row
        "has ever received a gradient".  Rows that never did have g = m = v = 0: Adam (no weight decay) and the gradient norm skip
        them exactly (csrc/optim.hip), and zero_grad() only clears the rows the last step touched.  The flags follow the ids the
        model logs per forward (and, under data parallelism, the ids every rank sent); anything the trainer cannot account for
        -- an overflowed log, a gradient accumulated outside the flat buffer, a loaded optimiser state -- raises the flags."""
        self._lazy = None
        fn = getattr(self.model, 'sparse_grad_rows', None)
        if not enabled or fn is None or self.S is None or self.wd != 0.0:
            return
        names = [n for n in fn() if n in self.names]
        if len(names) != 1:
            return
        i = self.names.index(names[0])
        p = self.params[i]
        if p.dim() != 2 or p.shape[1] % 4 != 0:
            return
        dev = p.device
        self._lazy = {'name': names[0], 'lo': self.offsets[i], 'rows': p.shape[0], 'rowlen': p.shape[1],
                      'flags': torch.zeros(p.shape[0], device=dev, dtype=torch.uint8),
                      'dirty': None, 'dirty_n': 0, 'all_dirty': True}  # rows whose gradient is non-zero right now

    def _lazy_arg(self):
        z = self._lazy
        return None if z is None else (z['flags'], z['lo'], z['rows'], z['rowlen'])

    def _lazy_mark(self, ids):
        """Rows `ids` (int64, any shape; -1 = blank) carry a gradient this step.  ids None: unknown -> every row from now on."""
        z = self._lazy
        if z is None:
            return
        if ids is None:
            z['flags'].fill_(1)
            z['all_dirty'] = True
            return
        ids = ids.reshape(-1).clamp_min(0)
        z['flags'].index_fill_(0, ids, 1)
        if z['all_dirty']:
            return
        n = ids.numel()
        if z['dirty'] is None or z['dirty_n'] + n > z['dirty'].numel():
            if z['dirty'] is not None and z['dirty_n'] > 0 or torch.cuda.is_current_stream_capturing():
                z['all_dirty'] = True  # no room (and no allocation inside a capture): the next zero_grad clears the whole table
                return
            z['dirty'] = torch.zeros(max(4 * n, 1024), device=ids.device, dtype=torch.int64)
            z['dirty_n'] = 0
        z['dirty'][z['dirty_n']:z['dirty_n'] + n].copy_(ids)
        z['dirty_n'] += n

    def _shadow_view(self, p):
        i = next(k for k, q in enumerate(self.params) if q is p)
        o = self.offsets[i]
        return self.S[o:o + p.numel()].view(p.shape)

    def _attach_shadows(self):
        """Hand the model views of the flat bf16 shadow for every weight its MFMA kernels read in bf16: the tower's
        matrices and each head listed by `model.head_shadow_targets()` (BERT: to_logits; ART-V: the 51,584-way to_logits).
        A weight that is frozen (not in the flat buffer) keeps the model's own cast-on-change copy."""
        m = self.model
        self._tower_shadow_attached = False
        tw = getattr(m, 'transformer', None)
        if tw is not None and hasattr(tw, 'attach_shadow') and all(p.requires_grad for p in tw._matrix_params()):
            tw.attach_shadow([self._shadow_view(p) for p in tw._matrix_params()])
            self._tower_shadow_attached = True
        if hasattr(m, 'attach_head_shadow') and hasattr(m, 'head_shadow_targets'):
            for lin in m.head_shadow_targets():
                if lin.weight.requires_grad:
                    m.attach_head_shadow(lin, self._shadow_view(lin.weight))

    def _check_bindings(self):
        """`p.data` / `p.grad` must still be the views into P / G the kernels update: `model.zero_grad()` (set_to_none),
        `load_state_dict(assign=True)` or `.to()` would silently detach them.  Re-bind gradients, refuse moved params."""
        for p, o in zip(self.params, self.offsets):
            if p.data_ptr() != self.P.data_ptr() + 4 * o:
                raise RuntimeError('a parameter no longer lives in FlatTrainer.P (moved by .to() / load_state_dict(assign=True)); '
                                   'rebuild the trainer after moving the model')
            if p.grad is None or p.grad.data_ptr() != self.G.data_ptr() + 4 * o:
                stray = p.grad
                p.grad = self.G[o:o + p.numel()].view(p.shape)
                if stray is not None:  # gradients were accumulated into a detached tensor this step: fold them in
                    p.grad.add_(stray)
                    if self._lazy is not None and o == self._lazy['lo']:
                        self._lazy_mark(None)  # rows the id log does not know about

    # --------------------------------------------------------------------------------------------- checkpoint
    def state_dict(self):
        """Optimiser state in torch.optim.Adam's layout (train.py:202-203, 352, 387 save / restore `optimizer`): per
        parameter exp_avg / exp_avg_sq / step, keyed by position in `self.names` order (recorded under 'names'), plus the
        front-end's (seed, forward count) so a resumed run continues its mask / warp stream instead of replaying it."""
        state = {}
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            n = p.numel()
            state[i] = {'step': torch.tensor(float(self.step_count)), 'exp_avg': self.M[o:o + n].view(p.shape).clone(),
                        'exp_avg_sq': self.V[o:o + n].view(p.shape).clone()}
        group = {'lr': self.lr, 'betas': self.betas, 'eps': self.eps, 'weight_decay': self.wd, 'amsgrad': False,
                 'params': list(range(len(self.params)))}
        out = {'state': state, 'param_groups': [group], 'names': list(self.names)}
        fe = getattr(self.model, 'frontend', None)
        if fe is not None and hasattr(fe, 'state_dict'):
            out['frontend'] = fe.state_dict()
        return out

    def load_state_dict(self, sd):
        """Accepts this class's own state_dict and a plain torch.optim.Adam one (the reference's checkpoints,
        train.py:352): without a 'names' entry the indices are torch's -- position in the requires_grad-filtered
        `model.parameters()` order the reference hands to Adam (utils_train.py:167-172) -- and are mapped through the
        parameter names to the flat layout; every entry's shape is checked, so moments can never land on another layer."""
        if 'names' in sd:
            names = list(sd['names'])
        else:
            names = [n for n, p in self.model.named_parameters() if p.requires_grad]
        if sorted(names) != sorted(self.names):
            raise ValueError('optimizer state was saved for a different set of parameters: '
                             f'{sorted(set(names) ^ set(self.names))[:6]} ...')
        at = {n: i for i, n in enumerate(self.names)}
        steps, loaded = set(), []
        for j, name in enumerate(names):
            st = sd['state'].get(j, sd['state'].get(str(j)))
            if st is None:
                continue
            i = at[name]
            p, o, n = self.params[i], self.offsets[i], self.params[i].numel()
            for k in ('exp_avg', 'exp_avg_sq'):
                if tuple(st[k].shape) != tuple(p.shape):
                    raise ValueError(f'optimizer state entry {j} ({name}): {k} has shape {tuple(st[k].shape)}, the parameter '
                                     f'{tuple(p.shape)}')
            loaded.append((o, n, st))
            steps.add(int(float(st['step'])))
        if len(steps) > 1:
            raise ValueError('per-parameter step counts differ')
        for o, n, st in loaded:
            self.M[o:o + n].copy_(st['exp_avg'].reshape(-1))
            self.V[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
        if steps:
            self.step_count = steps.pop()
            if self._step_dev is not None:
                self._step_dev.fill_(float(self.step_count))
        g = sd['param_groups'][0]
        self.lr, self.betas, self.eps, self.wd = g['lr'], tuple(g['betas']), g['eps'], g['weight_decay']
        fe = getattr(self.model, 'frontend', None)
        if fe is not None and 'frontend' in sd and hasattr(fe, 'load_state_dict'):
            fe.load_state_dict(sd['frontend'])
        z = getattr(self, '_lazy', None)
        if z is not None:  # rows with loaded moments must keep decaying
            lo, hi = z['lo'], z['lo'] + z['rows'] * z['rowlen']
            live = (self.M[lo:hi].view(z['rows'], -1).abs().amax(1) > 0) | (self.V[lo:hi].view(z['rows'], -1).amax(1) > 0)
            z['flags'].copy_(live.to(torch.uint8) | z['flags'])

    def refresh_shadows(self):
        """Call after writing parameters behind the trainer's back (model.load_state_dict): re-cast the bf16 shadow."""
        if self.S is not None:
            ops.cast_bf16(self.P, self.S)
            self._attach_shadows()

    # ---------------------------------------------------------------------------------------------
    def zero_grad(self):
        z = self._lazy
        if z is not None:
            # Table rows a backward wrote that no step() recorded as dirty (a step skipped on a non-finite loss; a backward whose forward
            # ran before the previous zero_grad()) must not survive as stale rows behind unflagged rows: the model raises
            # `table_grad_pending` from a hook on the backward itself and step() lowers it once the rows are recorded, so a raised flag
            # here means "unaccounted rows" -> clear the whole table.  Two zero_grad() calls in a row raise nothing (no 152-MB fill).
            # A model without the flag: any zero_grad() that does not follow a step() clears the whole table.
            pending = getattr(self.model, 'table_grad_pending', None)
            if pending is None:
                pending = not getattr(self, '_stepped_since_zero', True)
            if pending or getattr(self.model, '_text_id_overflow', False):
                z['all_dirty'] = True
        self._stepped_since_zero = False
        if z is None or z['all_dirty']:
            self.G.zero_()
        else:  # everything but the lazy table, and of the table only the rows the last step left a gradient in
            lo, hi = z['lo'], z['lo'] + z['rows'] * z['rowlen']
            self.G[:lo].zero_()
            self.G[hi:].zero_()
            if z['dirty_n']:
                self.G[lo:hi].view(z['rows'], z['rowlen']).index_fill_(0, z['dirty'][:z['dirty_n']], 0.0)
        if z is not None:
            z['all_dirty'], z['dirty_n'] = False, 0
            if hasattr(self.model, 'table_grad_pending'):
                self.model.table_grad_pending = False
        reset = getattr(self.model, 'reset_sparse_grad_rows', None)
        if reset is not None:
            reset()

    def _sparse_ranges(self):
        """Flat ranges of the tables whose gradient is exchanged row-wise (see _exchange_sparse): excluded from the all-reduce."""
        fn = getattr(self.model, 'sparse_grad_rows', None)
        if not self.sparse_tables or fn is None:
            return []
        out = []
        for name, ids in fn().items():
            # ids None = the model has no (complete) list of the rows this step touched -- no forward logged yet, or its
            # log overflowed under gradient accumulation: that table goes through the dense all-reduce like any other
            if name in self.names and ids is not None:
                i = self.names.index(name)
                out.append((self.offsets[i], self.offsets[i] + _round_up(self.params[i].numel(), ALIGN)))
        return sorted(out)

    def _send(self, lo, hi):
        """All-reduce G[lo:hi] (SUM) on the side stream, in messages of at most bucket_elems."""
        if (self.world == 1 and not self.force_exchange) or hi <= lo or not self.exchange_enabled:
            return
        pieces, at = [], lo
        for a, b in self._sparse_ranges():  # cut the row-wise exchanged tables out of [lo, hi)
            if b <= at or a >= hi:
                continue
            if a > at:
                pieces.append((at, a))
            at = max(at, b)
        if at < hi:
            pieces.append((at, hi))
        if self._comm_stream is not None:
            self._comm_stream.wait_stream(torch.cuda.current_stream())
        for plo, phi in pieces:
            for s in range(plo, phi, self.bucket_elems):
                e = min(s + self.bucket_elems, phi)
                if self._comm_stream is not None:
                    with torch.cuda.stream(self._comm_stream):
                        self._reduce_message(s, e)
                else:
                    self._reduce_message(s, e)

    def _reduce_message(self, s, e):
        """G[s:e] <- sum over ranks (on the current stream)."""
        if self.exchange == 'allreduce' or self.world == 1 or (e - s) % self.world != 0:  # (a group of one has no peers to send shards to)
            self._works.append(dist.all_reduce(self.G[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
            return
        # direct reduce-scatter + all-gather: fixed buffers (capturable), every peer link busy at once
        n, w = e - s, self.world
        if self._direct_buf is None or self._direct_buf.numel() < n + n // w:
            self._direct_buf = torch.empty(self.bucket_elems + self.bucket_elems // w + w, device=self.G.device, dtype=torch.float32)
        recv = self._direct_buf[:n].view(w, n // w)
        shard = self._direct_buf[n:n + n // w]
        msg = self.G[s:e]
        nccl = dist.get_backend(self.pg) == 'nccl'
        if nccl:
            dist.all_to_all_single(recv.view(-1), msg, group=self.pg)
        else:  # gloo (CPU tests) has no all_to_all: the same exchange as point-to-point sends
            me, parts = self._rank(), msg.view(w, n // w)
            recv[me].copy_(parts[me])
            reqs = [dist.P2POp(dist.isend, parts[j].contiguous(), j, self.pg) for j in range(w) if j != me] + \
                   [dist.P2POp(dist.irecv, recv[j], j, self.pg) for j in range(w) if j != me]
            for r in dist.batch_isend_irecv(reqs):
                r.wait()
        torch.sum(recv, dim=0, out=shard)  # rank order: the same bits on every rank
        if nccl:
            dist.all_gather_into_tensor(msg, shard, group=self.pg)
        else:
            dist.all_gather(list(msg.view(w, n // w).unbind(0)), shard, group=self.pg)

    def _gather(self, t):
        """[n, ...] on every rank -> [world * n, ...] (rank-major)."""
        out = torch.empty((self.world * t.shape[0], ) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        if dist.get_backend(self.pg) == 'nccl':
            dist.all_gather_into_tensor(out, t, group=self.pg)
        else:
            dist.all_gather(list(out.chunk(self.world, 0)), t, group=self.pg)
        return out

    def _exchange_sparse(self):
        """Row-wise gradient exchange of the tables `model.sparse_grad_rows()` names (BERT: the 49,472 x 768 text embedding,
        152 MB dense, of which a step touches at most B * 64 rows).  The table is the LAST gradient of the backward, so its
        dense all-reduce cannot overlap anything: instead every rank gathers (row id, gradient row) of the rows ITS batch
        touched -- ids sorted, repeats blanked, fixed shapes, so the step stays capturable -- and adds the other ranks' rows
        to its own table gradient: the same sum, ~1 MB per rank on the wire instead of 152 MB."""
        fn = getattr(self.model, 'sparse_grad_rows', None)
        if not self.sparse_tables or fn is None or not self.exchange_enabled:
            return
        rank = self._rank()
        for name, ids in fn().items():
            if name not in self.names or ids is None:
                continue  # None: the table was not cut out of the dense all-reduce (_sparse_ranges)
            W = self.params[self.names.index(name)].grad  # [V, E] view into G
            uid, rows = self.pack_rows(W, ids)
            all_ids = self._gather(uid)
            self.merge_rows(W, all_ids, self._gather(rows), rank, uid.shape[0])
            if self._lazy is not None and self._lazy['name'] == name:
                self._lazy_mark(all_ids)  # the other ranks' rows now carry a gradient here too

    def _rank(self):
        return dist.get_rank(self.pg)

    @staticmethod
    def pack_rows(W, ids):
        """This rank's message: (ids sorted, repeats blanked to -1) and the matching gradient rows (zero where blanked)."""
        ids = ids.reshape(-1)
        if W.is_cuda and ids.numel() <= 4096 and W.dtype == torch.float32 and W.is_contiguous():  # two launches, no allocator traffic beyond the outputs
            uid = torch.empty_like(ids)
            rows = torch.empty(ids.numel(), W.shape[1], device=W.device, dtype=W.dtype)
            _lib.call('mmvid_rows_pack', ops._p(W), W.shape[0], W.shape[1], ops._p(ids.contiguous()), ids.numel(), ops._p(uid), ops._p(rows),
                      ops._stream())
            return uid, rows
        srt, _ = torch.sort(ids)
        first = torch.ones_like(srt, dtype=torch.bool)
        first[1:] = srt[1:] != srt[:-1]
        uid = torch.where(first, srt, torch.full_like(srt, -1))
        rows = W.index_select(0, srt) * first.unsqueeze(1).to(W.dtype)
        return uid, rows

    @staticmethod
    def merge_rows(W, all_ids, all_rows, rank, n):
        """Add every OTHER rank's rows (rank-major gathered messages of n rows each) to this rank's table gradient."""
        if W.is_cuda and W.dtype == torch.float32 and W.is_contiguous():  # one launch per peer, in rank order: a fixed summation order
            for r in range(all_ids.shape[0] // n):
                if r != rank:
                    _lib.call('mmvid_rows_merge', ops._p(W), W.shape[0], W.shape[1], ops._p(all_ids[r * n:(r + 1) * n]),
                              ops._p(all_rows[r * n:(r + 1) * n]), n, ops._stream())
            return
        own = torch.zeros(all_ids.shape[0], dtype=torch.bool, device=all_ids.device)
        own[rank * n:(rank + 1) * n] = True
        valid = (all_ids >= 0) & ~own
        W.index_add_(0, all_ids.clamp_min(0), all_rows * valid.unsqueeze(1).to(W.dtype))

    def layers_done(self, first_layer):
        """Tower backward callback: gradients of layers >= first_layer (and everything after them in the flat
        buffer: later layers, heads) are final -> put them on the wire while the backward continues."""
        lo = self._layer_start.get(first_layer)
        if lo is None or lo >= self._sent_from:
            return
        self._send(lo, self._sent_from)
        self._sent_from = lo

    def after_failed_capture(self):
        """A step that was being captured raised (e.g. a collective the runtime cannot capture).  The communication stream joined that
        capture through an event and HIP leaves it in capture mode when the origin's capture is torn down: every later launch on it fails
        with 'operation not permitted when stream is capturing'.  Take a fresh stream and forget the half-finished exchange."""
        if self._comm_stream is not None:
            self._comm_stream = torch.cuda.Stream(device=self.G.device)
        self._works = []
        self._sent_from = self.numel

    def allreduce_grads(self):
        """Send whatever is still local (embedding tables, or everything when no callback fired) and wait."""
        if self.world > 1 or self.force_exchange:
            self._send(0, self._sent_from)
            if self._comm_stream is not None:
                with torch.cuda.stream(self._comm_stream):  # after the dense messages, on the same stream
                    self._exchange_sparse()
            else:
                self._exchange_sparse()
            for w in self._works:
                w.wait()
            self._works = []
            if self._comm_stream is not None:
                torch.cuda.current_stream().wait_stream(self._comm_stream)
        self._sent_from = self.numel

    def step(self):
        """clip_grad_norm_(max_norm) + Adam (train.py:324-325) on the averaged gradients."""
        self._check_bindings()
        self.allreduce_grads()
        if self._lazy is not None:
            ids = self.model.sparse_grad_rows().get(self._lazy['name'])
            # a multi-rank step whose table went through the DENSE all-reduce (sparse_tables off) carries the other ranks' rows
            # too, and this rank's id log does not know them: every row counts from now on
            dense_exchange = (self.world > 1 or self.force_exchange) and self.exchange_enabled and not self.sparse_tables
            self._lazy_mark(None if dense_exchange else ids)
            if hasattr(self.model, 'table_grad_pending'):
                self.model.table_grad_pending = False  # (the rows are now recorded as dirty)
        self._stepped_since_zero = True
        self.step_count += 1
        gscale = 1.0 / self.world
        # the update itself is the HIP kernel; on a host tensor ops.adam_step raises (there is no CPU path)
        self._sq.zero_()
        lr_dev = None
        if self._step_dev is not None:
            if self.lr_schedule is not None:  # lr of THIS iteration from the count of finished steps, on the device
                sc = self.lr_schedule
                ops.lr_schedule(self._step_dev, 1, sc.min_lr, sc.max_lr, sc.warmup, sc.every, self._lr_dev)
                lr_dev = self._lr_dev
            ops.counter_add(self._step_dev, 1.0)
        ops.grad_sqnorm(self.G, self._sq, partials=self._sq_partials, lazy=self._lazy_arg())
        ops.adam_step(self.P, self.G, self.M, self.V, self.S, self.step_count, self.lr, self.betas, self.eps, self.wd,
                      self.max_norm, self._sq, gscale, step_dev=self._step_dev, lr_dev=lr_dev, lazy=self._lazy_arg())
        tw = getattr(self.model, 'transformer', None)
        if tw is not None and hasattr(tw, 'mark_shadow_fresh') and getattr(self, '_tower_shadow_attached', False):
            tw.mark_shadow_fresh()  # the Adam kernel wrote the attached bf16 views; an un-attached tower recasts itself

    def grad_norm(self):
        return float(self.G.norm()) / self.world


class GraphedStep:
    """One training step -- zero_grad, forward, backward, gradient exchange, clip + Adam -- as ONE hipGraph
    (torch.cuda.CUDAGraph).

    The step is several hundred kernel launches issued from Python and from the native layer loops; replayed as a graph
    it costs the host one call, so a slow or contended host (8 ranks sharing a 16-core quota) can no longer stall the GPU.
    `fn(**inputs)` returns the loss and must be capture-safe: device work only, fixed shapes; the stochastic front-end
    draws on the device from a device step counter, so replays need no host input at all.

    With torch.distributed the bucketed all-reduces are captured too: the tower backward returns to the host every few
    layers DURING CAPTURE, `FlatTrainer.layers_done` enqueues the finished range on the communication stream (which
    joins the capture through an event), and the replayed graph carries the same fork / join dependencies -- RCCL
    kernels overlapping the remaining backward.  If the runtime refuses to capture a collective, the step falls back to
    eager launches (`graph is None`, `capture_error` says why).

        step = GraphedStep(trainer, fn, example_inputs)      # runs `warmup` real steps, then captures
        loss = step()                                        # replay on the static inputs
        loss = step(text=..., frames=...)                    # copy new data into the static inputs, then replay
    """

    def __init__(self, trainer, fn, example_inputs, warmup=2):
        self.trainer, self.fn = trainer, fn
        self.static = {k: v.clone() for k, v in example_inputs.items()}
        self.graph, self.capture_error = None, None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up on a side stream, as graph capture requires (real training steps)
            for _ in range(warmup):
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        saved = (trainer.step_count, trainer._step_dev.clone(), self._frontend_steps())
        graph = torch.cuda.CUDAGraph()
        # with a process group alive, its watchdog thread polls events while we capture: only THIS thread's calls may be
        # policed (the default "global" mode turns the watchdog's hipEventQuery into a capture violation)
        mode = 'thread_local' if (dist.is_available() and dist.is_initialized()) else 'global'
        try:
            with torch.cuda.graph(graph, capture_error_mode=mode):
                self.loss = self._step()
            self.graph = graph
        except Exception as e:  # e.g. a collective that cannot be captured: keep training, eagerly
            self.capture_error = f'{type(e).__name__}: {e}'.splitlines()[0][:200]
            try:
                torch.cuda.synchronize()
            except Exception:  # (the failed capture's own error may surface once more here)
                pass
            trainer.after_failed_capture()
        # the capture pass enqueued nothing: it was not a step
        trainer.step_count = saved[0]
        trainer._step_dev.copy_(saved[1])
        self._frontend_steps(saved[2])

    def _frontend_steps(self, restore=None):
        fe = getattr(self.trainer.model, 'frontend', None)
        if fe is None or fe.step is None:
            return None
        if restore is not None:
            fe.step.copy_(restore)
            return None
        return fe.step.clone()

    def _step(self):
        self.trainer.zero_grad()
        loss = self.fn(**self.static)
        loss.backward()
        self.trainer.step()
        return loss.detach()

    def __call__(self, **inputs):
        for k, v in inputs.items():
            self.static[k].copy_(v, non_blocking=True)
        if self.graph is None:
            return self._step()
        self.graph.replay()
        self.trainer.step_count += 1
        return self.loss


_CAPTURABLE = {}  # (world size, device) -> the answer of collectives_capturable, once per process


def collectives_capturable(device):
    """Can this runtime capture a collective of the current backend into a hipGraph?  Asked BEFORE the training step is captured, on a
    throw-away process group: a collective that fails inside a capture leaves the group's internal streams in capture mode (HIP does not
    release the streams that joined a capture that is torn down), and every later EAGER collective on that group fails with 'operation
    not permitted when stream is capturing' -- the eager fall-back would be dead too.  gloo copies through the host: never capturable.
    All ranks return the same answer."""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    if dist.get_backend() != 'nccl':
        return False
    key = (dist.get_world_size(), str(device))
    if key in _CAPTURABLE:
        return _CAPTURABLE[key]

    def vote(ok):  # every rank learns whether EVERY rank got this far (the default group: untouched by the trial)
        flag = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    pg = None
    try:
        pg = dist.new_group(backend='nccl')
    except Exception:
        pg = None
    ok = vote(pg is not None)  # (no rank enters a collective of the trial group unless all of them have it)
    if ok:
        t = torch.zeros(64, device=device)
        try:
            dist.all_reduce(t, group=pg)  # (the communicator is built by the first, eager, collective)
            torch.cuda.synchronize()
        except Exception:
            ok = False
        ok = vote(ok)
    if ok:
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                dist.all_reduce(t, group=pg)
            g.replay()
            torch.cuda.synchronize()
        except Exception:
            ok = False
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
        ok = vote(ok)
    # (the trial group is kept: destroying a group whose collective sits in a captured graph aborts the process on this runtime; the
    #  answer is cached per process, so exactly one extra communicator exists however often steps are captured)
    _CAPTURABLE[key] = ok
    return ok


def backward_order(params):
    """Flat-buffer order = REVERSE of the order gradients become final in the backward, so that reversed buckets
    (heads + last layers first) can be sent while earlier layers are still being differentiated."""
    def key(item):
        n = item[0]
        if n.startswith('transformer.transformer.resblocks.'):
            return (1, int(n.split('.')[3]))
        if n.startswith('to_logits'):
            return (2, 0)
        return (0, 0)  # embeddings / positional tables: final only when the whole backward is done

    return sorted(params, key=key)


def broadcast_parameters(model, src=0, group=None):
    """DDP construction broadcast (train.py:32): parameters and buffers from rank 0."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)
