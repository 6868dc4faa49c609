import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from test_dyn_mask_head_gpu import _bf16_chain_reference, _case
from torch.nn import functional as F
from pctrans_amd import dynamic_mask_head as dmh
N, Q, H, W, target = 1, 4, 16, 128, (8, 64)
mf, ref, prm = _case(N, Q, H, W, seed=5)
mf, ref, prm = mf.float().cuda(), ref.float().cuda(), prm.float().cuda()
want = F.interpolate(_bf16_chain_reference(mf, ref, prm).bfloat16(), size=(2 * H, 2 * W), mode=os.environ.get("DBG_MODE","bilinear"), **({} if os.environ.get("DBG_MODE") else {"align_corners": False})).float()
for k in ("mfma", "fused"):
    up, am = dmh.dynamic_mask_head_forward(mf, ref.transpose(0, 1), prm.transpose(0, 1).contiguous(), 4, True, target, out_dtype=torch.bfloat16, kernel=k)
    d = (up.float() - want).abs()
    bad = d > (want.abs() * 2.0 ** -6 + 3e-2)
    print(k, "bad fraction", float(bad.float().mean()))
    print(" per query", bad.float().mean(dim=(0, 2, 3)).tolist())
    print(" per out row (q0)", [round(x, 2) for x in bad[0, 0].float().mean(dim=1).tolist()])
    print(" per out col/16 (q0)", [round(x, 2) for x in bad[0, 0].float().mean(dim=0).view(16, 16).mean(1).tolist()])
    print(" sample", up[0, 0, 4, :8].float().tolist(), want[0, 0, 4, :8].tolist())
if os.environ.get("DBG_MODE") == "nearest":
    L = _bf16_chain_reference(mf, ref, prm).bfloat16().float()      # [N, Q, H, W]
    up, am = dmh.dynamic_mask_head_forward(mf, ref.transpose(0, 1), prm.transpose(0, 1).contiguous(), 4, True, target, out_dtype=torch.bfloat16, kernel="fused")
    Fz = up.float()[:, :, ::2, ::2]      # par 0 rows, even cols -> pixel (y, x)
    for (q, y, x) in [(0, 2, 0), (0, 2, 1), (0, 2, 8), (0, 5, 17), (1, 2, 0), (2, 9, 40)]:
        v = float(Fz[0, q, y, x])
        d = (L[0] - v).abs()
        idx = torch.nonzero(d < 2e-3)
        print("fused[q=%d,y=%d,x=%d]=%.4f ref there %.4f; ref matches at" % (q, y, x, v, float(L[0, q, y, x])), idx[:6].tolist())
