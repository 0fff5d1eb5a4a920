"""HIP-graph replay of the inference forward (the launch-bound case: batch 1 at 384x1280 is ~300 kernels of ~10 us each and
the Python enqueue takes as long as the GPU needs).  ``GraphedDepth`` captures ``depth_net(rgb)`` once into a HIP graph
(torch.cuda.CUDAGraph: every kernel of libmte_hip.so is launched on the capturing stream, the library keeps no host state
between launches in eval mode) and replays it for every new frame copied into the static input buffer.

Measured on MI355X, bf16, 1 x 384x1280: 4.5 ms eager -> 3.8 ms replay per frame; at batch >= 4 the forward is GPU-bound and
the graph gives nothing.

``GraphedTrainStep`` captures the whole optimisation step -- zero_grad, forward (Dropout2d draws included), the fused loss
launch, backward with its weight-gradient side stream, fused Adam and the weight-pack prefetch: ~1000 launches -- into two HIP
graphs, one per outcome of the reference's whole-batch flip draw (models/SfmModel.py:87-96), which stays a host-side
``random.random()`` taken before each replay.  The eager step needs ~18 ms of Python per 30 ms of GPU work; a replay needs
well under 1 ms, so the launch path stops being within reach of the GPU time."""
import torch


class GraphedDepth:
    def __init__(self, depth_net, example_rgb, warmup=3):
        from .. import kernels as K
        K._require_gpu(example_rgb)
        if depth_net.training:
            raise ValueError("capture the network in eval mode (dropout / flip decisions are host-side state)")
        self.net = depth_net
        self.rgb = example_rgb.detach().clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                        # builds the weight packs / workspaces outside the capture
                self.net(self.rgb)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        K.begin_graph_capture()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.out = self.net(self.rgb)
        K.end_graph_capture()

    def __call__(self, rgb):
        """rgb: same shape as the example -> the network's output dict (tensors are overwritten by the next call)."""
        if tuple(rgb.shape) != tuple(self.rgb.shape):
            raise ValueError("graph captured for {}, got {}".format(tuple(self.rgb.shape), tuple(rgb.shape)))
        self.rgb.copy_(rgb)
        self.graph.replay()
        return self.out


class GraphedTrainStep:
    """``step = GraphedTrainStep(model, optimizer, batch); loss = step(batch)`` == ``optimizer.zero_grad(); out = model(batch);
    out['loss'].backward(); optimizer.step()`` of the reference's inner loop (trainers/common_trainer.py:119-125).

    * The batch tensors are static buffers: a new batch is copied into them (same shapes).
    * Randomness: the flip decision is drawn on the host per step with the model's own ``draw_flip`` (python RNG, as the
      reference) and selects one of two captured graphs; Dropout2d factors are drawn by torch's generator inside the graph
      (graph-safe Philox offsets: fresh draws every replay).
    * Single-process only: the bucketed RCCL all-reduce is issued from grad-ready callbacks and stays on the eager path.
    * If capture fails the object reports it (``.graphed is False``) and every call runs the eager step in this process.
    * Construction leaves the training state untouched: the warm-up runs real optimisation steps on the example batch (kernel
      attributes, workspaces and weight packs must exist before a capture), so parameters, Adam moments and step count, module
      buffers (BatchNorm running statistics of the sparse branch) and the python / torch RNG states are snapshotted first and
      restored afterwards, and the kernel-ready weight packs are rebuilt from the restored parameters."""

    def __init__(self, model, optimizer, example_batch, warmup=2):
        import random
        from .. import kernels as K
        self.model, self.opt = model, optimizer
        self.batch = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
        self.graphs, self.outs, self.error = {}, {}, None
        if getattr(optimizer, "reducer", None) is not None and getattr(optimizer.reducer, "active", False):
            self.error = "gradient all-reduce active: eager step"
            return
        if not model.training:
            raise ValueError("capture the model in training mode")
        rng_state = random.getstate()
        snap = self._snapshot()
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):                     # weight packs, workspaces, hipFuncSetAttribute: all outside the capture
                    for flip in (False, True):
                        self._eager(flip)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            pool = None
            for flip in (False, True):
                g = torch.cuda.CUDAGraph()
                K.begin_graph_capture()
                with torch.cuda.graph(g, pool=pool):
                    self.outs[flip] = self._eager(flip)
                pool = g.pool()
                self.graphs[flip] = g
                K.end_graph_capture()
        except Exception as e:                              # stay usable: eager in this process (never re-exec a GPU process)
            self.error = "%s: %s" % (type(e).__name__, e)
            self.graphs, self.outs = {}, {}
            try:
                K.join_side_stream()
            except Exception:
                pass
            torch.cuda.synchronize()
        finally:
            self.model._pinned_flip = None
            random.setstate(rng_state)                      # warm-up / capture must not consume the training run's flip draws
            self._restore(snap)

    def _snapshot(self):
        opt = self.opt
        flat = getattr(opt, "flatp", None)
        snap = {"torch_rng": torch.get_rng_state(), "cuda_rng": torch.cuda.get_rng_state(),
                "buffers": [(b, b.detach().clone()) for b in self.model.buffers()]}
        if flat is not None:
            snap.update(flat=flat.flat.clone(), m=opt.exp_avg.clone(), v=opt.exp_avg_sq.clone(), steps=opt.steps)
        else:
            import copy
            snap.update(params=[(p, p.detach().clone()) for g in opt.param_groups for p in g["params"]],
                        opt_state=copy.deepcopy(opt.state_dict()))
        return snap

    def _restore(self, snap):
        from .. import kernels as K
        torch.cuda.synchronize()
        opt = self.opt
        with torch.no_grad():
            if "flat" in snap:
                opt.flatp.flat.copy_(snap["flat"])
                opt.exp_avg.copy_(snap["m"])
                opt.exp_avg_sq.copy_(snap["v"])
                opt.steps = snap["steps"]
            else:
                for p, v in snap["params"]:
                    p.copy_(v)
                opt.load_state_dict(snap["opt_state"])
            for b, v in snap["buffers"]:
                b.copy_(v)
        torch.set_rng_state(snap["torch_rng"])
        torch.cuda.set_rng_state(snap["cuda_rng"])
        # a replay does not run the python that re-packs stale weights: rebuild every pack from the restored parameters now
        K.bump_weights_epoch()
        K.prefetch_weight_packs()
        K.join_side_stream()
        torch.cuda.synchronize()

    @property
    def graphed(self):
        return len(self.graphs) == 2

    def _eager(self, flip):
        self.model._pinned_flip = bool(flip)
        self.opt.zero_grad()
        out = self.model(self.batch)
        out["loss"].backward()
        self.opt.step()
        return {"loss": out["loss"].detach(), "metrics": {k: v.detach() for k, v in out.get("metrics", {}).items()}}

    def __call__(self, batch=None):
        if batch is not None and batch is not self.batch:
            for k, v in batch.items():
                if torch.is_tensor(v) and k in self.batch:
                    if v.data_ptr() != self.batch[k].data_ptr():
                        self.batch[k].copy_(v, non_blocking=True)
        flip = self.model.draw_flip()
        if not self.graphed:
            try:
                return self._eager(flip)
            finally:
                self.model._pinned_flip = None
        self.opt.advance()                                  # step count, lr, bias corrections -> device memory the Adam node reads
        self.graphs[flip].replay()
        return self.outs[flip]
