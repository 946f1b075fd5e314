"""HIP-event stage timer used by bench.py: records events on torch's current stream (the stream
every kernel of the engine is launched on) around each stage / kernel of a forward pass, without
any synchronisation inside the timed region.  `summary()` synchronises once at the end."""
import contextlib
from collections import OrderedDict

import torch


class StageTimer:
    def __init__(self):
        self._ranges = []        # (name, start, end)
        self._layer_sets = []    # (name, [events], interval names | None)
        self._pool = []          # events created (and recorded once, so that their hipEvent_t exists) ahead of time

    def reserve(self, n):
        """Create n events NOW (outside any timed region): creating an event and its first record cost
        CPU time and a GPU marker each; inside a timed loop that starves the launch queue."""
        for _ in range(n):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._pool.append(ev)

    def _event(self):
        if self._pool:
            return self._pool.pop()
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    @contextlib.contextmanager
    def range(self, name):
        s, e = self._event(), self._event()
        s.record()
        try:
            yield
        finally:
            e.record()
            self._ranges.append((name, s, e))

    def layer_events(self, name, n=12, names=None):
        """n events whose hipEvent_t handles exist (torch creates them lazily on first record).
        names: the n-1 interval labels (default: summary()'s layer_names)."""
        evs = [self._event() for _ in range(n)]
        self._layer_sets.append((name, evs, names))
        return evs

    def summary(self, layer_names=None):
        """-> OrderedDict name -> {"ms": total, "calls": n}."""
        torch.cuda.synchronize()
        out = OrderedDict()

        def add(name, ms):
            d = out.setdefault(name, {"ms": 0.0, "calls": 0})
            d["ms"] += ms
            d["calls"] += 1
        for name, s, e in self._ranges:
            add(name, s.elapsed_time(e))
        for name, evs, names in self._layer_sets:
            names = names or layer_names
            for i in range(len(evs) - 1):
                lname = names[i] if names else str(i)
                add(f"{name}/{lname}", evs[i].elapsed_time(evs[i + 1]))
        return out

    def reset(self):
        self._ranges.clear()
        self._layer_sets.clear()
        self._pool.clear()


def stage(timer, name):
    return timer.range(name) if timer is not None else contextlib.nullcontext()
