import sys, os, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from msda_cases import make_case
from pctrans_amd import MultiScaleDeformableAttention as MSDA, _lib
lib = _lib.lib()
P2 = [(16, 16), (32, 32), (64, 64), (128, 128)]
S = sum(h * w for h, w in P2)
c = make_case(seed=79, N=2, M=8, D=16, Lq=S, P=4, shapes=P2, model_like=True, px_sigma=2.0)
go = np.random.RandomState(179).standard_normal((2, S, 128)).astype(np.float32)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
args = [dev(c["value"]), dev(c["shapes"]), dev(c["starts"]), dev(c["loc"]), dev(c["attn"]), dev(go), 64]
kern = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lib.pct_msda_set_bwd_kernel_choice(kern)
eager = MSDA.ms_deform_attn_backward(*args); torch.cuda.synchronize()
scale = float(eager[0].abs().max())
def err(o): return float((o[0] - eager[0]).abs().max()) / scale
for _ in range(5): e2 = MSDA.ms_deform_attn_backward(*args)
torch.cuda.synchronize(); print("A eager x5 no sync:", err(e2))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    MSDA.ms_deform_attn_backward(*args)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = MSDA.ms_deform_attn_backward(*args)
for i in range(3):
    g.replay(); torch.cuda.synchronize(); print("B replay no fill", i, err(out))
for i in range(2):
    out[0].fill_(float("nan")); g.replay(); torch.cuda.synchronize(); print("C replay, grad_value filled NaN", i, err(out))
for i in range(2):
    out[0].fill_(1.0); g.replay(); torch.cuda.synchronize(); print("D replay, grad_value filled 1.0", i, err(out))
for i in range(2):
    out[1].fill_(float("nan")); out[2].fill_(float("nan")); g.replay(); torch.cuda.synchronize(); print("E replay, grad_loc/attn filled NaN", i, err(out))
