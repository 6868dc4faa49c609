import sys, os, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from msda_cases import make_case
from pctrans_amd import MultiScaleDeformableAttention as MSDA, _lib
lib = _lib.lib()
P2 = [(16, 16), (32, 32), (64, 64), (128, 128)]
S = sum(h * w for h, w in P2)
sig = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
c = make_case(seed=79, N=2, M=8, D=16, Lq=S, P=4, shapes=P2, model_like=True, px_sigma=sig)
go = np.random.RandomState(179).standard_normal((2, S, 128)).astype(np.float32)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
args = [dev(c["value"]), dev(c["shapes"]), dev(c["starts"]), dev(c["loc"]), dev(c["attn"]), dev(go), 64]
lib.pct_msda_set_bwd_kernel_choice(3)
eager = MSDA.ms_deform_attn_backward(*args); torch.cuda.synchronize()
scale = float(eager[0].abs().max())
for i in range(6):
    e2 = MSDA.ms_deform_attn_backward(*args); torch.cuda.synchronize()
    print("eager", i, float((e2[0] - eager[0]).abs().max()) / scale, bool(torch.equal(e2[1], eager[1])), bool(torch.equal(e2[2], eager[2])))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    MSDA.ms_deform_attn_backward(*args)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = MSDA.ms_deform_attn_backward(*args)
for i in range(8):
    for t in out: t.fill_(float("nan"))
    g.replay(); torch.cuda.synchronize()
    d = (out[0] - eager[0]).abs()
    print("replay", i, float(d.max()) / scale, int((d > 1e-3 * scale).sum()), bool(torch.equal(out[1], eager[1])), bool(torch.equal(out[2], eager[2])))
d = (out[0] - eager[0]).abs().view(2, S, 8, 16)
bad = d > 1e-3 * scale
print("bad per channel", bad.sum((0, 1, 2)).tolist())
print("bad per head", bad.sum((0, 1, 3)).tolist())
print("bad per image", bad.sum((1, 2, 3)).tolist())
st = [0, 256, 1280, 5376, S]
print("bad per level", [int(bad[:, st[i]:st[i + 1]].sum()) for i in range(4)])
r = (out[0] / eager[0])[bad.view_as(out[0])]
print("ratio out/eager on bad: median %.3f min %.3f max %.3f" % (float(r.median()), float(r.min()), float(r.max())))
