import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from msda_cases import make_case
from pctrans_amd import MultiScaleDeformableAttention as MSDA
FULL = dict(N=8, M=8, D=16, P=4, shapes=[(16, 16), (32, 32), (64, 64), (128, 128)])
c = make_case(seed=81, model_like=True, Lq=21760, **FULL)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
v, loc, attn, sh, st = dev(c["value"]), dev(c["loc"]), dev(c["attn"]), dev(c["shapes"]), dev(c["starts"])
go = torch.randn(8, 21760, 128, device="cuda", generator=torch.Generator("cuda").manual_seed(5))
for i in range(40):
    out = MSDA.ms_deform_attn_forward(v, sh, st, loc, attn, 64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, sh, st, loc, attn, go, 64)
    lhs = (go.double() * out.double()).sum().item()
    tol = 1e-7 * (go.double().abs() * out.double().abs()).sum().item()
    print("lhs %.4f  d_gv %.5f  d_ga %.5f  tol %.4f  old_tol %.4f" % (lhs, (gv.double()*v.double()).sum().item()-lhs, (ga.double()*attn.double()).sum().item()-lhs, tol, 1e-5*abs(lhs)+1e-2))
