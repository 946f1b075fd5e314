"""Whole-forward hipGraph: `eval.py:213-222` calls `CascadeMVSNet.forward` once per reference view with the same
shapes every time, and one forward is ~50 kernel launches of 5-300 us each.  Captured once into a hipGraph
(torch.cuda.CUDAGraph drives hipStreamBeginCapture on ROCm) the launches cost the host one hipGraphLaunch, and the
short kernels of the U-Net's bottom (conv3..conv9: ~20 us each) run back to back instead of at the host's launch rate.

Everything the engine launches goes to torch's current stream with caller-owned buffers, no allocation, no
synchronisation and no host read-back inside the library (include/casmvs.h), so a forward is capturable as is; the
per-kernel one-time set-up (LDS opt-in, occupancy queries, weight packing, workspaces, cached per-level constants)
happens in the warm-up calls before the capture.
"""
import torch

from . import streams


def _uses_f16(model):
    """Does a forward of this model launch kernels with f16 / bf16 matrix instructions (the split layer modes)?"""
    return any(getattr(m, "conv0_mode", "f32") != "f32" or getattr(m, "ci_mode", "f32") != "f32" or getattr(m, "tail_mode", "f32") != "f32" or
               (getattr(m, "s2_mode", None) or "f32") != "f32" for m in model.modules())


class GraphedForward:
    """model(imgs, proj_mats, init_depth_min, depth_interval) as one hipGraph replay.

    The graph is captured for the shapes of the example inputs; `__call__` copies new inputs into the static input
    buffers, replays, and returns the STATIC output tensors (overwritten by the next call: clone what must survive).
    init_depth_min / depth_interval: python floats are baked into the graph (they select cached constant tensors);
    (B,1) tensors are copied into static buffers like the images."""

    def __init__(self, model, imgs, proj_mats, init_depth_min, depth_interval, warmup=2):
        if model.training:
            raise RuntimeError("GraphedForward captures the inference engine: call model.eval() first")
        if not imgs.is_cuda:
            raise RuntimeError("GraphedForward needs device-resident example inputs")
        self.model = model
        self.imgs = imgs.clone()
        self.proj_mats = proj_mats.clone()
        self.init_depth_min = init_depth_min.clone() if isinstance(init_depth_min, torch.Tensor) else init_depth_min
        self.depth_interval = depth_interval.clone() if isinstance(depth_interval, torch.Tensor) else depth_interval
        timer = model.timer
        model.set_timer(None)  # events cannot be recorded into a capture
        side = torch.cuda.Stream(device=imgs.device)
        side.wait_stream(torch.cuda.current_stream(imgs.device))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                model(self.imgs, self.proj_mats, self.init_depth_min, self.depth_interval)
        torch.cuda.current_stream(imgs.device).wait_stream(side)
        torch.cuda.synchronize(imgs.device)
        self.f16 = _uses_f16(model)
        streams.reset(imgs.device)   # the device is idle: nothing of the warm-up (another stream) can overlap the capture stream's launches
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = model(self.imgs, self.proj_mats, self.init_depth_min, self.depth_interval)
        model.set_timer(timer)

    def refresh_weights(self):
        """After the model's parameters changed (an optimiser step, load_state_dict): re-pack the folded weight images IN
        PLACE, so that the next replay - which reads them through the pointers captured in the graph - sees the new
        weights.  No re-capture."""
        dev = self.imgs.device
        self.model.feature.packed_layers(dev)
        for l in range(self.model.levels):
            getattr(self.model, f"cost_reg_{l}").packed_layers(dev)

    def __call__(self, imgs=None, proj_mats=None, init_depth_min=None, depth_interval=None):
        if imgs is not None and imgs.data_ptr() != self.imgs.data_ptr():
            self.imgs.copy_(imgs, non_blocking=True)
        if proj_mats is not None and proj_mats.data_ptr() != self.proj_mats.data_ptr():
            self.proj_mats.copy_(proj_mats, non_blocking=True)
        for name, new in (("init_depth_min", init_depth_min), ("depth_interval", depth_interval)):
            cur = getattr(self, name)
            if new is None:
                continue
            if isinstance(cur, torch.Tensor):
                cur.copy_(new.reshape(cur.shape), non_blocking=True)
            elif float(new) != float(cur):
                raise ValueError(f"GraphedForward: {name} = {new} differs from the captured constant {cur}; capture a new "
                                 "graph (python floats are baked in) or pass (B,1) tensors when the range varies")
        streams.note_launch(self.imgs.device, f16=self.f16)   # the replay's kernels run on the current stream: the cross-stream rule of streams.py
        self.graph.replay()
        return self.outputs


def shared_parameter_replica(model):
    """A copy of the module tree - own workspaces, own packed-weight images, own timers - whose Parameters and buffers ARE
    the source model's tensor objects (deepcopy with those objects pre-seeded in the memo)."""
    import copy
    memo = {id(t): t for t in list(model.parameters()) + list(model.buffers())}
    return copy.deepcopy(model, memo)


class ConcurrentForwards:
    """N captured forwards on N HIP streams, for throughput over INDEPENDENT reference views (eval.py:213 iterates them
    with no cross-iteration state).

    Why: one forward is a chain of ~50 dependent kernels; about a third of them (the bottom of the 3D U-Net, the
    level-2 layers, the softmax / hypothesis kernels) launch fewer workgroups than the chip holds, and every kernel has a
    drain tail.  A second, independent forward on another stream fills those holes: measured on the MI355X
    (tools/gpu_streams_probe.py) 2 streams x batch 2 = 684 depth maps/s against 627 for one stream (batch 4 on one
    stream: 663; 3 streams: no further gain).  Each stream owns a replica of the module TREE (its workspaces and packed
    weight images, ~10 MB) whose Parameters and buffers are the source model's own tensor objects
    (`shared_parameter_replica`): a weight update of the model is a weight update of every replica, and
    `refresh_weights()` re-packs the images in place for the captured graphs.

    `run(batches)` takes one (imgs, proj_mats) pair per stream (None = reuse the captured inputs), replays the graphs
    concurrently and returns the list of STATIC output dicts after making the caller's stream wait for all of them.

    MATRIX-INSTRUCTION TYPES ACROSS THE STREAMS.  Rounds 3-4 measured wrong float32 values in kernels that shared SIMDs with another stream's f16 /
    bf16 matrix instructions (800 of 800 replays for the Cout = 8 float32 layer kernel, 91 of 400 for the cost-volume kernel) and ran the replicas
    all-float32.  Round 5 found the cause - ONE instruction form, packed float32 with op_sel:[0,1,..] (casmvsnet_pl_amd/streams.py,
    tools/probes/pk_fma_opsel_repro.hip) - and the library is now assembled without it: with such a library (`casmvs_packed_opsel_safe()`) the
    replicas keep the model's own layer modes (split-f16) and every replay equals the single-stream forward bit for bit
    (tools/gpu_mixed_streams.py: 0 of 5 640 output tensors differ on 2-4 streams; 2 streams x batch 1: 990 depth maps/s vs 752 on one stream).
    `mixed_matrix_types=False` forces the all-float32 replicas (what a library built without the rewrite gets by default), True forces the model's
    modes."""

    def __init__(self, model, imgs, proj_mats, init_depth_min, depth_interval, n_streams=2, warmup=2, mixed_matrix_types=None):
        self.device = imgs.device
        if mixed_matrix_types is None:
            mixed_matrix_types = not streams.enabled()   # a library without the unsafe packed-float32 form: nothing to keep apart
        self.mixed_matrix_types = mixed_matrix_types = bool(mixed_matrix_types)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(n_streams)]
        self.forwards = []
        for st in self.streams:
            replica = shared_parameter_replica(model)
            if not mixed_matrix_types:
                for m in replica.modules():
                    if hasattr(m, "conv0_mode"):
                        m.conv0_mode = "f32"
                    if hasattr(m, "ci_mode"):
                        m.ci_mode = "f32"
                    if hasattr(m, "tail_mode"):
                        m.tail_mode = "f32"
                    if hasattr(m, "s2_mode"):   # an explicit "splitf16" on conv1 / conv3 would survive ci_mode = "f32" (s2_mode or ci_mode)
                        m.s2_mode = "f32"
            st.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(st), streams.stream_guard(not mixed_matrix_types):
                self.forwards.append(GraphedForward(replica, imgs, proj_mats, init_depth_min, depth_interval, warmup))
        torch.cuda.synchronize(self.device)

    def __len__(self):
        return len(self.forwards)

    def refresh_weights(self):
        """Re-pack every replica's weight images in place after the (shared) parameters changed."""
        for gf in self.forwards:
            gf.refresh_weights()

    def run(self, batches=None):
        cur = torch.cuda.current_stream(self.device)
        outs = []
        for i, (gf, st) in enumerate(zip(self.forwards, self.streams)):
            st.wait_stream(cur)   # inputs produced on the caller's stream are complete before the copy / replay
            with torch.cuda.stream(st), streams.stream_guard(not self.mixed_matrix_types):
                b = batches[i] if batches is not None else None
                outs.append(gf(*b) if b is not None else gf())
        for st in self.streams:
            cur.wait_stream(st)
        return outs
