#!/usr/bin/env python3
"""Kept measurement of the two TRAINING configurations of BASELINE.json on one MI355X (a record for profiles/, not the judged
metric): one optimiser step of the whole model -- ResNet-50 backbone, pixel decoder, transformer decoder with query contrast,
matcher on the device, criterion with deep supervision, backward, AdamW -- at the reference's per-GPU batch.

    configs[2]  configs/CVPPP/CVPPP-PCTrans.yaml: 2 x 448^2 crops (dataset_CVPPP.py:106), 3 encoder levels, 100 queries
    configs[3]  configs/BBBC/BBBC-PCTrans.yaml:   2 x 512^2 crops (dataset_BBBC.py:111), 3 encoder levels, 300 queries

Reported per configuration: wall time per step without any profiler (HIP events around whole steps + host clock), kernel
launches and device->host copies per step (torch profiler on ONE extra step, its own timing discarded), summed device time
of that step's kernels.  MIOpen's naive solvers are excluded from the search as in bench.py.  Synthetic data."""
import json
import os
import sys
import time

os.environ.setdefault("MIOPEN_FIND_MODE", "1")
os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "0")
os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD", "0")
os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pctrans_amd.arch import maskformer as mfm  # noqa: E402
from pctrans_amd.arch.resnet import ResNet  # noqa: E402
from pctrans_amd.config import get_cfg  # noqa: E402

CONFIGS = {
    "configs[2] CVPPP-PCTrans.yaml": dict(dataset="CVPPP", N=2, H=448, W=448, Q=100, G=20),
    "configs[3] BBBC-PCTrans.yaml": dict(dataset="BBBC", N=2, H=512, W=512, Q=300, G=60),
}


def build(c):
    import random
    random.seed(0)
    torch.manual_seed(0)
    cfg = get_cfg(num_queries=c["Q"], norm="BN", sem_norm="BN", dataset=c["dataset"])
    model = mfm.MaskFormer(**mfm.MaskFormer.from_config(cfg, ResNet(50, 3, norm="BN"))).cuda().train()
    N, H, W, G = c["N"], c["H"], c["W"], c["G"]
    vol = torch.randn(N, 3, H, W, device="cuda")
    g = torch.Generator().manual_seed(1)
    yy, xx = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
    targets = []
    for b in range(N):
        cy, cx = torch.randint(20, H - 20, (G,), generator=g), torch.randint(20, W - 20, (G,), generator=g)
        r = torch.randint(6, 16, (G,), generator=g)
        masks = torch.stack([(((yy - int(y)) ** 2 + (xx - int(x)) ** 2) <= int(q) ** 2).float() for y, x, q in zip(cy, cx, r)])
        centers = torch.stack([cx.float() / W, cy.float() / H], -1).view(G, 1, 2).cuda()
        targets.append({"masks": masks, "labels": torch.ones(G, dtype=torch.long, device="cuda"),
                        "fg_masks": (masks.sum(0) > 0).float(), "center_points": centers})
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.05)
    if c.get("graph_decoder"):
        from pctrans_amd.graph import graph_training_decoder
        graph_training_decoder(model, vol)
    if c.get("graph_front"):
        from pctrans_amd.graph import graph_training_front
        graph_training_front(model, vol)

    def step():
        losses = model(vol, targets, True)
        total = sum(v for v in losses.values() if torch.is_tensor(v))
        opt.zero_grad(set_to_none=True)
        total.backward()
        opt.step()
        return total
    return model, step


def main():
    out = {"device": torch.cuda.get_device_name(0), "note": __doc__.split("\n\n")[0], "configs": {}}
    configs = dict(CONFIGS)
    if "--graph-front" in sys.argv:
        configs.update({k + " + graphed backbone / pixel decoder": dict(v, graph_front=True) for k, v in CONFIGS.items()})
    if "--graph-decoder" in sys.argv:
        configs.update({k + " + graphed decoder core": dict(v, graph_decoder=True) for k, v in CONFIGS.items()})
    for name, c in configs.items():
        model, step = build(c)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        steps = 10
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            total = step()
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps * 1e3
        ev = e0.elapsed_time(e1) / steps
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
        launches, dev_us, d2h = 0, 0.0, 0
        top = {}
        for e in prof.events():
            if e.device_type == torch.autograd.DeviceType.CUDA:
                launches += 1
                dev_us += e.device_time if hasattr(e, "device_time") else e.cuda_time
                key = e.name[:60]
                top[key] = top.get(key, 0.0) + (e.device_time if hasattr(e, "device_time") else e.cuda_time)
                if "Memcpy DtoH" in e.name or "DtoH" in e.name:
                    d2h += 1
        rec = dict(c, step_ms_wall=round(wall, 2), step_ms_hip_events=round(ev, 2), kernel_launches_per_step=launches,
                   device_to_host_copies_per_step=d2h, device_ms_sum_of_kernels=round(dev_us / 1e3, 2),
                   loss=float(total.detach()), samples_per_s_per_gpu=round(c["N"] / (wall / 1e3), 1),
                   top_device_time_ms={k: round(v / 1e3, 2) for k, v in sorted(top.items(), key=lambda kv: -kv[1])[:12]})
        out["configs"][name] = rec
        print(name, json.dumps(rec), flush=True)
        if c.get("graph_front") or c.get("graph_decoder"):
            from pctrans_amd.graph import release_training_graphs
            release_training_graphs(model)                # graphs destroyed now, not by a later collector pass inside a replay
        del model, step, prof
        import gc
        gc.collect()
        torch.cuda.empty_cache()
    print("JSON " + json.dumps(out))


if __name__ == "__main__":
    main()
