"""Process-wide stream guard for the split-f16 kernels - OFF for a library whose device code carries no unsafe packed-float32 instruction.

The fault behind it (root cause found in round 5: tools/probes/pk_fma_opsel_repro.hip, profiles/r05_packed_opsel_fault_matrix.txt): on
gfx950, v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 whose low result takes src0's LOW half and a vector-register src1's HIGH half (op_sel:[0,1,..])
read that src1 half as zero in lanes 48-63 while another wave of the same SIMD issues f16 / bf16 matrix instructions.  That is what rounds 3-4 saw as
"float32 kernels return a few wrong values beside another stream's f16 kernels": the Cout = 8 float32-MFMA layer kernel (its epilogue has exactly one
such instruction: channel 7 of column tile 0, the element that was wrong in 171 of 200 rounds) and the packed-float32 plane loop of the LDS
cost-volume kernel (520 of them).  casmvsnet_pl_amd/build.py assembles the library with src0 / src1 of every such instruction exchanged (a clean form,
the same arithmetic bit for bit), `casmvs_packed_opsel_safe()` reports it, and with such a library this guard does nothing by default: its kernels run
beside each other on any number of streams (tools/gpu_mixed_streams.py: 0 of 5 640 output tensors of concurrent split-f16 forwards differ).

With a library built WITHOUT the rewrite (a plain `hipcc -c` of csrc/) the round-4 rule is applied:

    library kernels with f16 / bf16 matrix instructions never overlap library kernels of another stream.

Inside one stream kernels never overlap, so the engine's default launch (one stream, one hipGraph) never waits here.  When the library is driven from
several streams of one device, every launch goes through `launch_stream`: an f16-class launch first makes its stream wait for what the library has
queued on every other stream, and any launch first waits for the f16-class work queued on other streams (`Stream.wait_stream`: an event, no host
synchronisation).  Inside a hipGraph capture a cross-stream wait cannot be inserted: that combination raises.

What neither the rewrite nor the guard can see: kernels of OTHER code (torch, rocBLAS / MIOpen, RCCL) on other streams or processes of the same GPU that
contain the unsafe form - they are the victims then, whenever any f16 / bf16 matrix kernel (this library's or anyone's) shares their SIMDs.
`stream_guard(True / False)` forces the rule on / off.
"""
import contextlib
import ctypes

import os

import torch

_enabled = None   # None: decided by the loaded library on first use (casmvs_packed_opsel_safe); stream_guard() forces it
# device index -> {stream pointer: [torch stream, launches so far, f16-class launches so far]}
_streams = {}
# (device index, waiting stream pointer, other stream pointer) -> (launches, f16 launches) of the other stream already waited for
_seen = {}


@contextlib.contextmanager
def stream_guard(enabled):
    """Temporarily force the guard on / off (off with a library built without the packed-float32 rewrite: results are NOT reliable)."""
    global _enabled
    old, _enabled = _enabled, bool(enabled)
    try:
        yield
    finally:
        _enabled = old


def enabled():
    """Whether the cross-stream rule is being applied: forced by stream_guard, else on exactly when the loaded library may contain the unsafe form."""
    global _enabled
    if _enabled is None:
        # CASMVS_STREAM_GUARD=1 forces the rule on (and with it all-float32 replicas in graph.ConcurrentForwards), =0 off, whatever the library reports.
        # What casmvs_packed_opsel_safe() covers is THIS library's device code: float32 kernels of other code on other streams of the GPU (torch / MIOpen /
        # rocBLAS operators during a training step, RCCL) may contain the op_sel:[0,1,..] packed form and are not protected by it - INTEGRATION.md section 4.
        env = os.environ.get("CASMVS_STREAM_GUARD")
        if env in ("0", "1"):
            _enabled = env == "1"
        else:
            from . import _lib
            _enabled = not _lib.load().casmvs_packed_opsel_safe()
    return _enabled


def reset(device=None):
    """Forget the streams seen so far on `device` (all devices: None) - after torch.cuda.synchronize(device) nothing of theirs is in flight."""
    if device is None:
        _streams.clear()
        _seen.clear()
        return
    dev = torch.device(device).index
    dev = torch.cuda.current_device() if dev is None else dev
    _streams.pop(dev, None)
    for key in [k for k in _seen if k[0] == dev]:
        del _seen[key]


def note_launch(device, f16=False, stream=None):
    """Book a library launch (or a hipGraph replay of library kernels) on `stream` (default: torch's current stream of `device`) and insert the
    cross-stream waits the rule asks for.  -> the torch stream."""
    s = stream if stream is not None else torch.cuda.current_stream(device)
    if not enabled():
        return s
    dev = s.device.index if s.device.index is not None else torch.cuda.current_device()
    table = _streams.setdefault(dev, {})
    me = s.cuda_stream
    if len(table) > (me in table):   # other streams of this device have run library kernels
        for ptr, (other, launches, f16_launches) in list(table.items()):
            if ptr == me:
                continue
            done = _seen.get((dev, me, ptr), (0, 0))
            need = launches > done[0] if f16 else f16_launches > done[1]
            if need:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("casmvsnet_pl_amd: library kernels with f16 matrix instructions and kernels of another HIP stream may not overlap "
                                       "(casmvsnet_pl_amd/streams.py), and a hipGraph capture cannot wait for the other stream: capture mixed-type work on ONE "
                                       "stream, or select the float32 modes (conv0_mode / ci_mode / tail_mode = 'f32') for concurrent captures.  If the other "
                                       "stream's work is known to be complete (after torch.cuda.synchronize()), call casmvsnet_pl_amd.streams.reset(device) "
                                       "before the capture")
                if not other.query():   # work still queued / running there: an event wait (no host synchronisation)
                    s.wait_stream(other)
                _seen[(dev, me, ptr)] = (launches, f16_launches)
    entry = table.get(me)
    if entry is None:
        if len(table) > 64:   # a caller that keeps creating streams: drain the device once and start the bookkeeping over
            torch.cuda.synchronize(s.device)
            reset(s.device)
            table = _streams.setdefault(dev, {})
        table[me] = [s, 1, 1 if f16 else 0]
    else:
        entry[1] += 1
        entry[2] += 1 if f16 else 0
    return s


def launch_stream(t, f16=False):
    """The hipStream_t (ctypes.c_void_p) a library call on tensor `t`'s device goes to: torch's current stream, after the guard."""
    return ctypes.c_void_p(note_launch(t.device, f16).cuda_stream)
