import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from conftest import load_golden
import test_decoder_golden as T
for name, build in (("dec_full_decoder", None), ("dec_head_l4", None)):
    g = load_golden(name)
    if name == "dec_full_decoder":
        d = T._full_decoder("cuda")
        xs = [T._t(g["x%d" % i]).cuda() for i in range(3)]
        run = lambda: d(xs, None, T._t(g["mask_features"]).cuda())
    else:
        head = T._head("cuda")
        feats = {k: T._t(g["feat_" + k]).cuda() for k in T._PIX_CH}
        run = lambda: head(feats)[0]
    with torch.no_grad():
        o32 = run()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            o16 = run()
    ref = g["pred_masks"]; scale = max(1.0, float(np.abs(ref).max()))
    for tag, o in (("fp32", o32), ("bf16", o16)):
        a = o["pred_masks"].float().cpu().numpy()
        err = np.abs(a - ref)
        sign = np.mean((a > 0) == (ref > 0))
        rp = np.abs(o["reference_points"].float().cpu().numpy() - g["reference_points"]).max()
        print(name, tag, "scale %.2f max err/scale %.4f  p99.9 %.4f mean %.5f sign agree %.5f refpt err %.5f" % (scale, err.max()/scale, np.quantile(err, 0.999)/scale, err.mean()/scale, sign, rp))
        for i, au in enumerate(o["aux_outputs"]):
            a = au["pred_masks"].float().cpu().numpy(); e = np.abs(a - g["aux%d_pred_masks" % i])
            print("   aux%d max err/scale %.4f sign %.5f" % (i, e.max()/scale, np.mean((a > 0) == (g["aux%d_pred_masks" % i] > 0))))
