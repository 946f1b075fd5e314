"""Process-wide stream guard for the split-f16 kernels.

Measured on the MI355X (profiles/r03_mfma_coresidency*.txt, profiles/r04_coresidency_lib_victim.txt): while waves of one kernel
issue f16 / bf16 matrix instructions, some float32 kernels of ANOTHER stream that are co-resident on the same SIMDs return a few wrong values -
the Cout = 8 float32-MFMA layer kernel (conv16db_kernel<PX>: 171 of 200 rounds in a torch-free reproducer) and the packed-float32 plane loop of
the LDS cost-volume kernel; the other float32 layer kernels, the normalisation and softmax kernels were clean in the same experiments.  The cause
is not known (a stand-alone synthetic victim does not reproduce it: tools/probes/mfma_coresidency_repro.hip), so the rule is conservative:

    library kernels with f16 / bf16 matrix instructions never overlap library kernels of another stream.

Inside one stream kernels never overlap, so the engine's default launch (one stream, one hipGraph) is unaffected and never waits here.  When the
library is driven from several streams of one device, every launch goes through `launch_stream`: an f16-class launch first makes its stream wait for
what the library has queued on every other stream, and any launch first waits for the f16-class work queued on other streams (`Stream.wait_stream`:
an event, no host synchronisation).  All-float32 work on several streams - graph.ConcurrentForwards - is never serialised.  Inside a hipGraph
capture a cross-stream wait cannot be inserted: that combination raises.

What this cannot see: another process on the same GPU, kernels of other libraries (rocBLAS / MIOpen float32 GEMMs, RCCL) on other streams, and
callers of the C ABI that bypass this package.  `stream_guard(False)` switches it off (experiments: bench.py --unsafe-mixed-streams).
"""
import contextlib
import ctypes

import torch

_enabled = True
# device index -> {stream pointer: [torch stream, launches so far, f16-class launches so far]}
_streams = {}
# (device index, waiting stream pointer, other stream pointer) -> (launches, f16 launches) of the other stream already waited for
_seen = {}


@contextlib.contextmanager
def stream_guard(enabled):
    """Temporarily switch the guard (False: mixed-type work may overlap across streams - results are NOT reliable)."""
    global _enabled
    old, _enabled = _enabled, bool(enabled)
    try:
        yield
    finally:
        _enabled = old


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
    if not _enabled:
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
