#!/usr/bin/env python3
"""One-off diagnosis: what does the memset node look like that torch's capture records for the backward's
zeroing of grad_value, and what is wrong with it from the second replay on?  Needs a library built with
  -DPCT_EXPERIMENT_BUILD -DPCT_ZERO_BY_MEMSET   (zero_fill = hipMemsetAsync again)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from msda_cases import make_case
from pctrans_amd import MultiScaleDeformableAttention as MSDA, _lib
print("library:", _lib.lib().pct_build_info().decode())
hip = ctypes.CDLL("libamdhip64.so")
P2 = [(16, 16), (32, 32), (64, 64), (128, 128)]
S = sum(h * w for h, w in P2)
c = make_case(seed=79, N=2, M=8, D=16, Lq=S, P=4, shapes=P2, model_like=True, px_sigma=2.0)
go = np.random.RandomState(179).standard_normal((2, S, 128)).astype(np.float32)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
args = [dev(c["value"]), dev(c["shapes"]), dev(c["starts"]), dev(c["loc"]), dev(c["attn"]), dev(go), 64]
eager = MSDA.ms_deform_attn_backward(*args); torch.cuda.synchronize()
scale = float(eager[0].abs().max())
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    MSDA.ms_deform_attn_backward(*args)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph(keep_graph=True)
with torch.cuda.graph(g):
    out = MSDA.ms_deform_attn_backward(*args)
raw = g.raw_cuda_graph()
print("raw graph handle:", hex(raw), " grad_value ptr %#x bytes %d" % (out[0].data_ptr(), out[0].numel() * 4))


class MemsetParams(ctypes.Structure):
    _fields_ = [("dst", ctypes.c_void_p), ("elementSize", ctypes.c_uint), ("height", ctypes.c_size_t), ("pitch", ctypes.c_size_t),
                ("value", ctypes.c_uint), ("width", ctypes.c_size_t)]


def dump(tag):
    n = ctypes.c_size_t(0)
    hip.hipGraphGetNodes(ctypes.c_void_p(raw), None, ctypes.byref(n))
    nodes = (ctypes.c_void_p * n.value)()
    hip.hipGraphGetNodes(ctypes.c_void_p(raw), nodes, ctypes.byref(n))
    for i in range(n.value):
        t = ctypes.c_int(-1)
        hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t))
        line = "%s node %d type %d" % (tag, i, t.value)
        if t.value == 2:          # hipGraphNodeTypeMemset
            mp = MemsetParams()
            rc = hip.hipGraphMemsetNodeGetParams(ctypes.c_void_p(nodes[i]), ctypes.byref(mp))
            line += "  MEMSET rc=%d dst=%#x elementSize=%d width=%d height=%d pitch=%d value=%#x (width*elementSize=%d)" % (
                rc, mp.dst or 0, mp.elementSize, mp.width, mp.height, mp.pitch, mp.value, mp.width * mp.elementSize)
        print(line)


dump("captured")
g.instantiate()
for i in range(4):
    out[0].fill_(float("nan"))
    g.replay(); torch.cuda.synchronize()
    d = (out[0] - eager[0]).abs()
    bad = d > 1e-3 * scale
    flat = out[0].flatten()
    bi = torch.nonzero(bad.flatten())[:8].flatten().tolist()
    print("replay %d: max err %.4f x scale, bad elements %d of %d, first bad idx %s, their bit patterns %s" % (
        i, float(d.max()) / scale, int(bad.sum()), bad.numel(), bi, [hex(int(flat[j].view(torch.int32)) & 0xffffffff) for j in bi[:4]]))
    if bad.any():
        idx = torch.nonzero(bad.flatten()).flatten()
        print("   bad idx mod 4 histogram:", torch.bincount(idx % 4, minlength=4).tolist(), " mod 1024 of first: ", (idx[:6] % 1024).tolist(),
              " isnan there:", bool(torch.isnan(flat[idx[:64]]).any()))
dump("after replays")
