"""Optional per-launch timing shared by the kernel wrappers (bench.py's roofline leg): HIP events recorded on the
launch stream right before and after a kernel is enqueued.  Off by default; never synchronises while enabled."""
import torch

_TIMING = None


def kernel_timing(enable):
    """kernel_timing(True) starts collecting; kernel_timing(False) stops and returns [(name, milliseconds), ...]
    ((name, milliseconds, info) for launches that carry one: the MSDeformAttn forward reports the kernel id it ran)."""
    global _TIMING
    if enable:
        _TIMING = []
        return None
    rec, _TIMING = _TIMING or [], None
    out = []
    for name, e0, e1, info in rec:
        e1.synchronize()
        out.append((name, e0.elapsed_time(e1)) if info is None else (name, e0.elapsed_time(e1), info))
    return out


class timed:
    """Context manager around one launch; `t` is any tensor on the launch device."""

    def __init__(self, name, t, info=None):
        self.name, self.t, self.info = name, t, info          # info: callable evaluated right after the launch


    def __enter__(self):
        if _TIMING is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream(self.t.device))

    def __exit__(self, *exc):
        if _TIMING is not None and exc[0] is None:
            self.e1.record(torch.cuda.current_stream(self.t.device))
            _TIMING.append((self.name, self.e0, self.e1, self.info() if self.info is not None else None))
        return False
