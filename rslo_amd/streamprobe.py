"""Where a HIP stream's time goes, measured with HIP events only (no profiler attached).

`bench.py` switches this on for the last timed steps.  While it is on
  * every launching call into librslo_hip.so (capi.lib() hands out a proxy) is bracketed by two events on the stream it
    launches on: the call's busy time;
  * every explicit cross-stream join (`wait(name, stream, fn)`: streams.py leaf join, covariance-branch join, the operand
    pre-split, the structure plan of the prefetched batch) is bracketed the same way: how long the stream sat in that wait;
  * what is left between consecutive records of one stream is the launch boundary between dependent kernels plus whatever
    was issued there without a probe (torch's own element-wise kernels, joins the autograd engine inserts between the
    streams of a node and its consumer).
Two events cost host time and a marker packet each; `overhead_us()` measures an empty pair on the spot so that the gaps
can be read net of it.  Nothing here is on when the step is timed for the headline number.
"""
import threading
import time

import torch

_on = False
_lock = threading.Lock()
_records = []        # (stream id, kind, name, e0, e1), in issue order per issuing thread
_marks = []          # (stream id, label, event): step boundaries


def enabled():
    return _on


def start():
    global _on
    with _lock:
        del _records[:]
        del _marks[:]
    _on = True


def stop():
    global _on
    _on = False


def _sid(stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    return int(s.cuda_stream)


def mark(label, stream=None):
    """A named point on `stream` (default: the current one): the window the summary is taken over."""
    if not _on:
        return
    s = stream if stream is not None else torch.cuda.current_stream()
    ev = torch.cuda.Event(enable_timing=True)
    ev.record(s)
    if label == "begin":      # the host clock and the GPU timeline meet here: the stream is drained, so the event's time is now
        s.synchronize()
    with _lock:
        _marks.append((_sid(s), label, ev, time.perf_counter()))


class _Span:
    __slots__ = ("kind", "name", "stream", "e0", "h0")

    def __init__(self, kind, name, stream):
        self.kind, self.name, self.stream = kind, name, stream

    def __enter__(self):
        self.h0 = time.perf_counter()
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e0.record(self.stream)
        return self

    def __exit__(self, *exc):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(self.stream)
        with _lock:
            _records.append((_sid(self.stream), self.kind, self.name, self.e0, e1, self.h0))
        return False


def span(kind, name, stream=None):
    return _Span(kind, name, stream if stream is not None else torch.cuda.current_stream())


def wait(name, stream, fn):
    """Run the join `fn()` (a wait_stream / wait_event on `stream`); when probing, time how long `stream` waits in it."""
    if not _on:
        return fn()
    with _Span("wait", name, stream):
        return fn()


class LibProxy:
    """capi.lib() while probing: launching entry points (int status, last argument the stream) are timed on the current
    stream; everything else passes through."""

    def __init__(self, lib, signatures, points_only=None):
        import ctypes
        self._lib = lib
        self._points = set(points_only) if points_only else None      # sparse mode: ONE event + the host clock in front of these calls
        self._launching = {n for n, (res, args) in signatures.items()
                           if res is ctypes.c_int and args and args[-1] is ctypes.c_void_p
                           and not n.endswith(("_supported", "_layout")) and not n.startswith(("rslo_tuning", "rslo_peer_"))}
        self._cache = {}

    def __getattr__(self, name):
        fn = self._cache.get(name)
        if fn is None:
            raw = getattr(self._lib, name)
            if self._points is not None:
                if name in self._points:
                    def fn(*a, _raw=raw, _name=name):
                        if _on:
                            st = torch.cuda.current_stream()
                            ev = torch.cuda.Event(enable_timing=True)
                            h = time.perf_counter()
                            ev.record(st)
                            with _lock:
                                _records.append((_sid(st), "point", _name, ev, ev, h))
                        return _raw(*a)
                else:
                    fn = raw
            elif name in self._launching:
                def fn(*a, _raw=raw, _name=name):
                    if not _on:
                        return _raw(*a)
                    with _Span("call", _name, torch.cuda.current_stream()):
                        return _raw(*a)
            else:
                fn = raw
            self._cache[name] = fn
        return fn


def overhead_us(stream=None, n=64):
    """GPU-timeline cost of one empty event pair on an otherwise idle stream (microseconds)."""
    s = stream if stream is not None else torch.cuda.current_stream()
    s.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n)]
    for e in evs:
        e.record(s)
    s.synchronize()
    tot = sum(evs[2 * i].elapsed_time(evs[2 * i + 1]) for i in range(n))
    between = sum(evs[2 * i + 1].elapsed_time(evs[2 * i + 2]) for i in range(n - 1))
    return 1e3 * tot / n, 1e3 * between / max(n - 1, 1)


def summarize(train_stream, steps, top=12):
    """-> dict for the bench line: per-step busy / wait / gap milliseconds of the training stream (and busy time of the other
    streams), the largest gaps with the calls on either side, and the boundary count."""
    torch.cuda.synchronize()
    tid = _sid(train_stream)
    with _lock:
        recs = list(_records)
        marks = [m for m in _marks if m[0] == tid]
    out = {"steps": steps}
    if len(marks) < 2 or not recs:
        return out
    window_ms = marks[0][2].elapsed_time(marks[-1][2])
    by_stream = {}
    for r in recs:
        by_stream.setdefault(r[0], []).append(r)
    tr = by_stream.get(tid, [])
    # order the training stream's records on its own timeline (two threads issue: the step and the prefetch helper; the
    # helper launches on its own stream, so per-stream order == issue order, but sort by start offset to be safe)
    t0, h_begin = marks[0][2], marks[0][3]
    tr = sorted(((t0.elapsed_time(r[3]), r) for r in tr), key=lambda x: x[0])
    other = []          # (start, end, name) of the other streams' calls on the same timeline
    for sid, rs in by_stream.items():
        if sid != tid:
            for r in rs:
                if r[1] == "call":
                    st = t0.elapsed_time(r[3])
                    other.append((st, st + r[3].elapsed_time(r[4]), r[2]))
    busy = 0.0
    waits = {}
    gaps = []
    prev_end, prev_name = 0.0, "<step start>"
    n_calls = 0
    for start_ms, (sid, kind, name, e0, e1, h0) in tr:
        dur = e0.elapsed_time(e1)
        if start_ms < 0 or start_ms > window_ms:
            continue
        if kind == "wait":
            waits[name] = waits.get(name, 0.0) + dur
        else:
            busy += dur
            n_calls += 1
        gap = start_ms - prev_end
        if gap > 0:
            # how long before the GPU reached it the host had issued this call: ~0 = the stream was waiting for the HOST
            lead_us = 1e3 * (start_ms - 1e3 * (h0 - h_begin))
            busy_other = sum(min(e, start_ms) - max(s_, prev_end) for s_, e, _ in other if e > prev_end and s_ < start_ms)
            gaps.append((gap, prev_name, name, lead_us, busy_other))
        prev_end, prev_name = start_ms + dur, name
    tail = window_ms - prev_end
    if tail > 0:
        gaps.append((tail, prev_name, "<step end>", 0.0, 0.0))
    gap_total = sum(g[0] for g in gaps)
    per = 1.0 / max(steps, 1)
    pair_us, between_us = overhead_us(train_stream)
    out.update({
        "window_ms_per_step": round(window_ms * per, 3),
        "probed_call_ms_per_step": round(busy * per, 3),
        "probed_calls_per_step": int(n_calls * per),
        "wait_ms_per_step": {k: round(v * per, 3) for k, v in sorted(waits.items())},
        "between_probes_ms_per_step": round(gap_total * per, 3),
        "between_probes_note": "launch boundaries between dependent kernels + unprobed launches (torch element-wise / cat / "
                               "fill kernels, engine-inserted stream joins); includes the probe's own marker packets",
        "event_pair_us": round(pair_us, 2), "event_to_event_us": round(between_us, 2),
        "gaps_over_20us_per_step": round(sum(1 for g in gaps if g[0] > 0.02) * per, 1),
        "gaps_over_20us_ms_per_step": round(sum(g[0] for g in gaps if g[0] > 0.02) * per, 3),
        # gaps whose next call had been issued by the host less than 30 us before the GPU started it: the stream waited for
        # the issuing thread, not for the GPU (under the probe the host is slower than in the plain step)
        "gaps_host_starved_ms_per_step": round(sum(g[0] for g in gaps if g[0] > 0.02 and g[3] < 30.0) * per, 3),
        "largest_gaps": [{"us": round(1e3 * g, 1), "after": a, "before": b, "host_issued_before_gpu_start_us": round(ld, 1),
                          "other_streams_busy_us": round(1e3 * bo, 1)}
                         for g, a, b, ld, bo in sorted(gaps, key=lambda x: -x[0])[:top]],
    })
    others = {}
    for sid, rs in by_stream.items():
        if sid == tid:
            continue
        names = {}
        for r in rs:
            names[r[2]] = names.get(r[2], 0) + 1
        others[hex(sid)] = {"ms": round(sum(r[3].elapsed_time(r[4]) for r in rs if r[1] == "call") * per, 3),
                            "top_calls": sorted(names, key=lambda n: -names[n])[:3]}
    out["other_streams_probed_call_ms_per_step"] = others
    return out


def lag_profile(train_stream, n_bins=24):
    """Sparse mode: for the recorded points of the training stream, GPU time of the point minus the host time it was issued
    at (both zeroed at the drained `begin` mark), as (position in the window, lag ms) bins: where inside a step the issuing
    threads' lead over the GPU collapses."""
    torch.cuda.synchronize()
    tid = _sid(train_stream)
    with _lock:
        recs = [r for r in _records if r[0] == tid and r[1] == "point"]
        marks = [m for m in _marks if m[0] == tid]
    if len(marks) < 2 or not recs:
        return None
    t0, h0 = marks[0][2], marks[0][3]
    pts = [(t0.elapsed_time(r[3]), t0.elapsed_time(r[3]) - 1e3 * (r[5] - h0), r[2]) for r in recs]
    lags = sorted(p[1] for p in pts)
    n = len(lags)
    return {"points": n, "lag_ms_min": round(lags[0], 3), "lag_ms_p10": round(lags[n // 10], 3), "lag_ms_median": round(lags[n // 2], 3),
            "lag_ms_p90": round(lags[(9 * n) // 10], 3), "points_with_lag_under_50us": sum(1 for v in lags if v < 0.05),
            "points_with_lag_under_200us": sum(1 for v in lags if v < 0.2)}
