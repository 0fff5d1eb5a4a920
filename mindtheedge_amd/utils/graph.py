"""HIP-graph replay of the inference forward (the launch-bound case: batch 1 at 384x1280 is ~300 kernels of ~10 us each and
the Python enqueue takes as long as the GPU needs).  ``GraphedDepth`` captures ``depth_net(rgb)`` once into a HIP graph
(torch.cuda.CUDAGraph: every kernel of libmte_hip.so is launched on the capturing stream, the library keeps no host state
between launches in eval mode) and replays it for every new frame copied into the static input buffer.

Measured on MI355X, bf16, 1 x 384x1280: 4.5 ms eager -> 3.8 ms replay per frame; at batch >= 4 the forward is GPU-bound and
the graph gives nothing.  Training is not captured: its step is GPU-bound (host enqueue 15 ms per 31 ms step)."""
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

    def __call__(self, rgb):
        """rgb: same shape as the example -> the network's output dict (tensors are overwritten by the next call)."""
        if tuple(rgb.shape) != tuple(self.rgb.shape):
            raise ValueError("graph captured for {}, got {}".format(tuple(self.rgb.shape), tuple(rgb.shape)))
        self.rgb.copy_(rgb)
        self.graph.replay()
        return self.out
