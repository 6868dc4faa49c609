"""Optional per-launch timing shared by the kernel wrappers (bench.py's roofline leg): HIP events recorded on the
launch stream right before and after a kernel is enqueued.  Off by default; never synchronises while enabled."""
import torch

_TIMING = None
_POOL = []            # events created AND recorded once before a timed region (reserve): taken before new ones are made


def kernel_timing(enable, recycle=False):
    """kernel_timing(True) starts collecting; kernel_timing(False) stops and returns [(name, milliseconds), ...]
    ((name, milliseconds, info) for launches that carry one: the MSDeformAttn forward reports the kernel id it ran).
    `recycle` (with False): the events just read go back to the pool `timed` draws from, so a caller that times block after
    block keeps a bounded number of live events (bench.py's settle blocks: thousands of live timing events made
    `rocprofv3 --pmc` crash around the script)."""
    global _TIMING
    if enable:
        _TIMING = []
        return None
    rec, _TIMING = _TIMING or [], None
    out = []
    for name, e0, e1, info in rec:
        e1.synchronize()
        out.append((name, e0.elapsed_time(e1)) if info is None else (name, e0.elapsed_time(e1), info))
        if recycle:
            _POOL.append(e0)
            _POOL.append(e1)
    return out


def reserve(n, device=None):
    """Create n timing events and record each once now.  The runtime allocates an event's profiling signal at its first record,
    in pools that grow in steps; a timed region that records hundreds of fresh events can hit such a growth (bench.py, round 4:
    one step of 446 ms among steps of 91.7, two steps into the region that starts recording ~80 events per step).  Events
    reserved here are handed out by `timed` before any new one is created."""
    stream = torch.cuda.current_stream(device)
    fresh = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    for e in fresh:
        e.record(stream)
    stream.synchronize()
    _POOL.extend(fresh)


def _event():
    return _POOL.pop() if _POOL else torch.cuda.Event(enable_timing=True)


def count():
    """Launches recorded so far (0 when timing is off): lets a caller attribute the records to its own phases."""
    return len(_TIMING) if _TIMING is not None else 0


class timed:
    """Context manager around one launch; `t` is any tensor on the launch device."""

    def __init__(self, name, t, info=None):
        self.name, self.t, self.info = name, t, info          # info: callable evaluated right after the launch


    def __enter__(self):
        if _TIMING is not None:
            self.e0 = _event()
            self.e1 = _event()
            self.e0.record(torch.cuda.current_stream(self.t.device))

    def __exit__(self, *exc):
        if _TIMING is not None and exc[0] is None:
            self.e1.record(torch.cuda.current_stream(self.t.device))
            _TIMING.append((self.name, self.e0, self.e1, self.info() if self.info is not None else None))
        return False
