"""Flat parameter/gradient storage, bucketed gradient all-reduce (RCCL over xGMI) and the fused Adam optimizer.

This is the MI355X-native replacement of the reference's dead Horovod path (``hvd.DistributedOptimizer`` in
packnet_sfm/trainers/horovod_trainer.py:53-55: average every parameter gradient across ranks, overlapped with
backward, waited in ``optimizer.step()``) and of ``torch.optim.Adam`` (models/model_wrapper.py:142-180).

One process per GPU.  Parameters live in ONE flat fp32 buffer (77 M elements = 308 MB) and gradients in another, laid
out in *reverse execution order* so that gradient readiness during backward sweeps the buffer front to back.  The
gradient buffer is cut into ~32 MB buckets; a post-accumulate hook counts ready tensors per bucket and launches one
asynchronous ``all_reduce`` per bucket on RCCL's stream the moment the bucket is complete, so communication of the
decoder/late-encoder gradients overlaps the remaining backward.  xGMI is point-to-point (7 links/GPU): a few large
messages amortise the ring's per-link latency better than 218 small ones, and the two huge tensors
(pack5.conv 151 MB, pack4.conv 38 MB) finish early in backward, so their all-reduces hide under the high-resolution
layers' backward.  ``step()`` waits for the outstanding collectives and runs one fused Adam kernel over the flat buffers.
"""
import os

import torch
import torch.distributed as dist


class FlatParameters:
    """Re-homes ``params`` into one flat fp32 buffer (+ one flat gradient buffer of the same layout)."""

    def __init__(self, params, reverse=True, align=64):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("no trainable parameters")
        order = list(reversed(params)) if reverse else list(params)
        # parameters that normally receive no gradient (PackNetSAN01's SAN fusion scalars without a LiDAR input) go to the tail:
        # in the first bucket they would keep it from ever completing during backward, and its all-reduce would run exposed
        order = [p for p in order if not getattr(p, '_mte_flat_tail', False)] + [p for p in order if getattr(p, '_mte_flat_tail', False)]
        dev = order[0].device
        offs, total = [], 0
        for p in order:
            offs.append(total)
            total += (p.numel() + align - 1) // align * align
        self.params, self.offsets, self.total, self.reverse = order, offs, total, reverse
        self.natural_params = params                       # the order the caller gave (torch.optim's state index space)
        self.offset_of = {id(p): o for p, o in zip(order, offs)}
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, o in zip(order, offs):
            view = self.flat[o:o + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad[o:o + p.numel()].view(p.shape)
        self.sink = None
        if dev.type == "cuda":
            # backward kernels store conv / GroupNorm parameter gradients straight into self.grad (no per-tensor add)
            from .. import kernels as K
            self.sink = K.GradSink()
            for p in order:
                self.sink.register(p, p.grad)
            K.set_grad_sink(self.sink)

    def zero_grad(self):
        if self.grad.is_cuda:
            from .. import kernels as K
            K.join_side_stream()
        self.grad.zero_()
        if self.sink is not None:
            self.sink.reset()                                # every view may take one in-place store again
        for p, o in zip(self.params, self.offsets):          # re-attach if something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)


class BucketedAllReduce:
    """Overlapped gradient averaging over a FlatParameters gradient buffer."""

    def __init__(self, flat, process_group=None, bucket_bytes=32 << 20, force=False, tail_bytes=2 << 20, message_bytes=32 << 20):
        self.flat, self.group = flat, process_group
        # A bucket never splits a tensor, so the 151 MB pack5.conv weight makes a 182 MB bucket: its all-reduce goes out as
        # <= message_bytes pieces (xGMI is point-to-point: a ring step moves message/world per link, and 32 MB pieces keep the
        # pipeline of RCCL's ring busy while letting later, smaller buckets interleave instead of queueing behind one giant
        # message).  SURVEY.md 7: "bucket must split tensors".
        self.message_elems = max(1, message_bytes // 4)
        self.launch_log = []        # per step: (bucket index, first element, elements, messages) in launch order
        self.exposed_events = None  # (start, end) device events around the wait in finish(): the exposed all-reduce time
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.buckets = []                                   # (start, end, n_tensors)
        cap = max(1, bucket_bytes // 4)
        start, count, self.bucket_of = 0, 0, {}
        for i, (p, o) in enumerate(zip(flat.params, flat.offsets)):
            end = flat.offsets[i + 1] if i + 1 < len(flat.params) else flat.total
            self.bucket_of[id(p)] = len(self.buckets)
            count += 1
            if end - start >= cap or i + 1 == len(flat.params):
                self.buckets.append((start, end, count))
                start, count = end, 0
        # The last bucket (the first layers of the encoder) completes only when backward ends, so its all-reduce is the one
        # that cannot hide: keep that exposed message small by cutting the final bucket where ~2 MB of gradients remain.
        if self.buckets and self.buckets[-1][2] > 1:
            s0, e0, n0 = self.buckets[-1]
            first = len(flat.params) - n0
            for j in range(first + 1, len(flat.params)):
                if (flat.total - flat.offsets[j]) * 4 <= tail_bytes and (flat.offsets[j] - s0) * 4 >= tail_bytes:
                    self.buckets[-1] = (s0, flat.offsets[j], j - first)
                    self.buckets.append((flat.offsets[j], e0, n0 - (j - first)))
                    for p in flat.params[j:]:
                        self.bucket_of[id(p)] = len(self.buckets) - 1
                    break
        self._ready = [0] * len(self.buckets)
        self._seen = set()          # a parameter may be announced by both the gradient sink and its autograd hook
        self._works = []
        self._hooks = []
        self._stream = None         # collectives are issued from here (see _launch)
        self.active = self.world > 1 or (force and dist.is_available() and dist.is_initialized())   # force: single-rank RCCL tests
        if self.active:
            for p in flat.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad_ready))
            if getattr(flat, "sink", None) is not None:
                flat.sink.on_ready = self._on_grad_ready      # gradients written in place never reach AccumulateGrad

    def _on_grad_ready(self, p):
        if id(p) in self._seen:
            return
        self._seen.add(id(p))
        b = self.bucket_of[id(p)]
        self._ready[b] += 1
        if self._ready[b] == self.buckets[b][2]:
            s, e, _ = self.buckets[b]
            self._launch(s, e, b)

    def _messages(self, s, e):
        """[start, end) of a bucket cut into all-reduce messages of at most message_elems elements"""
        out = []
        while s < e:
            n = min(self.message_elems, e - s)
            out.append((s, s + n))
            s += n
        return out

    def _launch(self, s, e, b=-1):
        msgs = self._messages(s, e)
        self.launch_log.append((b, s, e - s, len(msgs)))
        if not self.flat.grad.is_cuda:
            for ms, me in msgs:
                self._works.append(dist.all_reduce(self.flat.grad[ms:me], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        # The bucket's gradients were produced on the main stream AND on the weight-gradient side stream.  Order the
        # collective after both from a third stream, so that the main stream (the rest of backward) never stalls on
        # the side stream's lag at a bucket boundary.
        from .. import kernels as K
        if self._stream is None:
            self._stream = torch.cuda.Stream()
        self._stream.wait_stream(torch.cuda.current_stream())
        K.side_streams_wait_into(self._stream)
        with torch.cuda.stream(self._stream):
            for ms, me in msgs:
                self._works.append(dist.all_reduce(self.flat.grad[ms:me], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Wait for every launched collective; reduce buckets whose hooks never completed (unused parameters).
        Returns the factor that turns the summed gradient into the average (1/world)."""
        if self.flat.grad.is_cuda:
            from .. import kernels as K
            K.join_side_stream()
        if self.active:
            for b, (s, e, n) in enumerate(self.buckets):
                if self._ready[b] != n:
                    self._launch(s, e, b)
            timed = self.flat.grad.is_cuda
            if timed:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            for w in self._works:
                w.wait()
            if timed:
                ev1.record()                                # ev0 -> ev1 on the main stream = time the step waits for RCCL
                self.exposed_events = (ev0, ev1)
        self.last_launches = self.launch_log
        self._works, self._ready, self.launch_log = [], [0] * len(self.buckets), []
        self._seen.clear()
        return 1.0 / self.world

    def exposed_ms(self):
        """Device time the last finish() spent waiting for outstanding collectives (synchronises)."""
        if self.exposed_events is None:
            return 0.0
        self.exposed_events[1].synchronize()
        return float(self.exposed_events[0].elapsed_time(self.exposed_events[1]))

    def describe(self):
        """Static layout + the last step's launch order, for bench.py's JSON line."""
        return {"buckets_mb": [round((e - s) * 4 / 2**20, 2) for s, e, _ in self.buckets],
                "message_mb": round(self.message_elems * 4 / 2**20, 2),
                "messages_per_bucket": [len(self._messages(s, e)) for s, e, _ in self.buckets],
                "launch_order": [b for b, _, _, _ in getattr(self, "last_launches", [])],
                "launch_offsets_mb": [round(s * 4 / 2**20, 2) for _, s, _, _ in getattr(self, "last_launches", [])]}


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(lr, betas, eps, weight_decay=0) semantics as ONE HBM-bound kernel over the flat buffers
    (28 B/parameter: read p,g,m,v; write p,m,v).  ``param_groups`` keeps torch's scheduler API (StepLR) working."""

    def __init__(self, flat, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, reducer=None, name='Depth'):
        self.flatp = flat
        self.reducer = reducer
        super().__init__([{'params': flat.params, 'name': name}], dict(lr=lr, betas=betas, eps=eps))
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        self.steps = 0
        self.hyper = torch.zeros(3, dtype=torch.float32, device=flat.flat.device)    # lr, 1 - b1^t, sqrt(1 - b2^t): read by the kernel
        self._hyper_pinned = None

    def zero_grad(self, set_to_none=False):
        self.flatp.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        from .. import kernels as K
        if self.flatp.grad.is_cuda:
            K.join_side_stream()
        gscale = self.reducer.finish() if self.reducer is not None else 1.0
        if self.flatp.sink is not None:
            self.flatp.sink.end_step()                       # (a forward pass's gradient-sink suspension lasts until its gradients are consumed)
        g = self.param_groups[0]
        if self.flatp.grad.is_cuda:
            # step count and learning rate reach the kernel through device memory, so the launch is identical every step (HIP-graph
            # replay); under capture the host side of the step (advance) is the replaying caller's job
            if not torch.cuda.is_current_stream_capturing():
                self.advance()
            K.lib.mte_adam_step_dev(self.flatp.flat.data_ptr(), self.flatp.grad.data_ptr(), self.exp_avg.data_ptr(),
                                    self.exp_avg_sq.data_ptr(), self.flatp.flat.numel(), self.hyper.data_ptr(),
                                    float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), float(gscale),
                                    torch.cuda.current_stream().cuda_stream)
            K.bump_weights_epoch()
        else:
            self.steps += 1
            K.adam_step_flat(self.flatp.flat, self.flatp.grad, self.exp_avg, self.exp_avg_sq, self.steps, lr=g['lr'],
                             betas=g['betas'], eps=g['eps'], gscale=gscale)
        if self.flatp.grad.is_cuda:
            K.prefetch_weight_packs()                       # next step's kernel-ready weight packs, off the critical path
            if not torch.cuda.is_current_stream_capturing():
                K.check_device_errors()                     # a bounded inter-workgroup wait gave up somewhere: fail the step (one host-memory read)

    def advance(self):
        """Host side of one step: count it and hand the kernel its learning rate and bias corrections (same double-precision
        arithmetic as mte_adam_step) -- called by step(), or by a graph-replaying caller right before each replay."""
        import math
        g = self.param_groups[0]
        self.steps += 1
        b1, b2 = g['betas']
        vals = [float(g['lr']), 1.0 - math.pow(float(b1), self.steps), math.sqrt(1.0 - math.pow(float(b2), self.steps))]
        if self.hyper.is_cuda and not os.environ.get("MTE_ADAM_PAGEABLE_HYPER"):
            # a copy from pageable memory makes the host wait for the stream (every step: the GPU then idles while the host catches
            # up with the next forward pass); a ring of pinned slots keeps the upload asynchronous.  Each slot carries the event of
            # its last upload and is rewritten only after that copy has executed (with HIP-graph replay or an unthrottled loop the host
            # can run more than 16 steps ahead of the device; the wait is almost always already satisfied)
            if self._hyper_pinned is None:
                self._hyper_pinned = torch.empty((16, 3), dtype=torch.float32).pin_memory()
                self._hyper_events = [None] * 16
            i = self.steps % 16
            if self._hyper_events[i] is not None:
                self._hyper_events[i].synchronize()
            slot = self._hyper_pinned[i]
            slot[0], slot[1], slot[2] = vals
            self.hyper.copy_(slot, non_blocking=True)
            ev = self._hyper_events[i] or torch.cuda.Event()
            ev.record()
            self._hyper_events[i] = ev
        else:
            self.hyper.copy_(torch.tensor(vals, dtype=torch.float32), non_blocking=True)

    def set_index_space(self, names, local_names):
        """Number the optimizer state like the reference's ``torch.optim.Adam(depth_net.parameters())`` does.

        names: every parameter name of the reference network in ``depth_net.parameters()`` order -- frozen tensors and the
        sparse-branch (``mconvs.*``) tensors included, whether or not this build materialises them (reference
        models/model_wrapper.py:149-154 passes ALL of depth_net.parameters()).  local_names: {name: parameter} of the
        tensors that live in the flat buffers.  Without it the state is numbered by position in the flat parameter list."""
        self._index_names = list(names)
        self._local = {n: p for n, p in local_names.items() if id(p) in self.flatp.offset_of}

    def _space(self):
        """[(index-space name or None, parameter or None, flat offset)] in the reference's numbering."""
        if getattr(self, '_index_names', None) is None:
            return [(None, p, self.flatp.offset_of[id(p)]) for p in self.flatp.natural_params]
        out = []
        for n in self._index_names:
            p = self._local.get(n)
            out.append((n, p, self.flatp.offset_of[id(p)] if p is not None else -1))
        return out

    def state_dict(self):
        """torch.optim.Adam's layout ({'state': {i: {'step','exp_avg','exp_avg_sq'}}, 'param_groups': [...]}).  Indices are
        positions in the reference's ``depth_net.parameters()`` (see set_index_space): 'params' has one entry per reference
        parameter, state entries exist for the tensors this build trains -- exactly what the reference's own Adam holds, since
        its sparse-branch tensors never receive a gradient on the RGB-only path and so never get state either
        (models/model_checkpoint.py:71-81 stores optimizer.state_dict())."""
        space = self._space()
        state = {}
        if self.steps > 0:
            for i, (_, p, o) in enumerate(space):
                if p is None:
                    continue
                n = p.numel()
                state[i] = {'step': torch.tensor(float(self.steps)), 'exp_avg': self.exp_avg[o:o + n].view(p.shape).clone(),
                            'exp_avg_sq': self.exp_avg_sq[o:o + n].view(p.shape).clone()}
        g = self.param_groups[0]
        group = {'lr': g['lr'], 'betas': tuple(g['betas']), 'eps': g['eps'], 'weight_decay': 0.0, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'name': g.get('name', 'Depth'), 'params': list(range(len(space)))}
        for k in ('initial_lr',):
            if k in g:
                group[k] = g[k]
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        if 'state' not in sd:                                   # flat layout written by earlier builds of this package
            self.steps = sd['steps']
            self.exp_avg.copy_(sd['exp_avg'])
            self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        else:
            space = self._space()
            groups = sd['param_groups']
            idx = [i for g in groups if g.get('name', 'Depth') == 'Depth' for i in g['params']] or list(groups[0]['params'])
            if len(idx) != len(space):
                # a dictionary numbered over the trainable tensors only (earlier builds of this package)
                trainable = [(n, p, o) for n, p, o in space if p is not None]
                if len(idx) != len(trainable):
                    raise ValueError("optimizer state holds {} 'Depth' parameters; this network numbers {} ({} of them trained here)"
                                     .format(len(idx), len(space), len(trainable)))
                space = trainable
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            steps = 0
            for i, (name, p, o) in zip(idx, space):
                st = sd['state'].get(i)
                if st is None or p is None:                     # no state yet / a tensor this build does not hold (mconvs.*)
                    continue
                if tuple(st['exp_avg'].shape) != tuple(p.shape):
                    raise ValueError("optimizer state {} ({}) has shape {}, parameter has {}".format(
                        i, name, tuple(st['exp_avg'].shape), tuple(p.shape)))
                n = p.numel()
                self.exp_avg[o:o + n].copy_(st['exp_avg'].reshape(-1))
                self.exp_avg_sq[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
                steps = max(steps, int(float(st['step'])))
            self.steps = steps                                  # torch keeps a step per tensor; they advance together here
        for g, s_ in zip(self.param_groups, sd['param_groups']):
            g.update({k: v for k, v in s_.items() if k in ('lr', 'betas', 'eps', 'initial_lr')})


def reference_parameter_names(depth_net):
    """Parameter names of the REFERENCE PackNetSAN01 in ``parameters()`` order.  torch yields a module's OWN parameters before
    those of its sub-modules, so the fusion vectors registered last in the constructor (networks/depth/PackNetSAN01.py:209-210)
    come first: ``weight, bias, encoder.*, decoder.*, mconvs.*``.  A network built without the sparse branch gets the branch's
    names from a meta-device instance (no memory) appended where the reference has them -- at the end -- so optimizer indices
    agree with checkpoints written by the reference and between ``with_san=True`` / ``False`` builds of this package."""
    names = [n for n, _ in depth_net.named_parameters()]
    if getattr(depth_net, 'with_san', True) or not hasattr(depth_net, 'mconvs'):
        return names
    with torch.device('meta'):
        from ..networks.layers.minkowski_encoder import MinkowskiEncoder
        branch = ['mconvs.' + n for n, _ in MinkowskiEncoder([32, 64, 128, 256, 512], with_uncertainty=False).named_parameters()]
    return names + branch


def broadcast_parameters(flat, src=0, group=None):
    """Rank-0 weights to every rank (Horovod's broadcast_parameters equivalent) -- one message."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat.flat, src=src, group=group)
