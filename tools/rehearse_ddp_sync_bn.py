#!/usr/bin/env python3
"""Rehearsal of the reference's multi-GPU TRAINING launch on a box with ONE GPU: N ranks (default 2) share cuda:0 over gloo
(RCCL refuses two ranks on one device), each runs one training step of the whole MaskFormer through
pctrans_amd.parallel.make_parallel(..., norm_mode="sync_bn") -- connectomics/model/build.py:74-102: BatchNorm ->
SyncBatchNorm, DistributedDataParallel(find_unused_parameters=True) -- so that the SyncBatchNorm statistics exchange, the
DDP gradient buckets and the criterion's num_masks all-reduce all execute on device tensors.  Not a measurement: the ranks
time-share one GPU.

    python tools/rehearse_ddp_sync_bn.py [--ranks 2] [--out DIR]      (parent: starts the ranks with torch.distributed.run)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rank_main(args):
    import random
    import torch
    import torch.distributed as dist
    from pctrans_amd import parallel
    from pctrans_amd.arch import maskformer as mfm
    from pctrans_amd.arch.resnet import ResNet
    from pctrans_amd.config import get_cfg
    device, rank, local_rank, world = parallel.init_devices(distributed=True, backend="nccl", share_gpu=True, manual_seed=0)
    assert device.type == "cuda" and dist.get_backend() == "gloo" and world == args.ranks
    cfg = get_cfg(num_queries=12, norm="BN", sem_norm="BN", enc_layers=2, dec_layers=3, train_num_points=512, dataset="BBBC")
    model = mfm.MaskFormer(**mfm.MaskFormer.from_config(cfg, ResNet(18, 3, norm="BN")))      # same seed on every rank
    n_bn = sum(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) for m in model.modules())
    H = W = 128
    g = torch.Generator(device="cuda").manual_seed(100 + rank)                                # a different shard per rank
    vol = torch.randn(2, 3, H, W, device="cuda", generator=g)
    if args.graph_decoder:
        # the decoder's static-shape core from HIP graphs on every rank: norms converted first, capture, then DDP
        from pctrans_amd import graph
        model = parallel.convert_norms(model).to(device)
        graph.graph_training_decoder(model, vol)
        ddp = parallel.make_parallel(model, device, parallel="DDP", norm_mode=None)
        assert graph.has_graphed_decoder(ddp.module)
    else:
        ddp = parallel.make_parallel(model, device, parallel="DDP", norm_mode="sync_bn")
    net = ddp.module
    n_sync = sum(isinstance(m, torch.nn.SyncBatchNorm) for m in net.modules())
    assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel) and ddp.find_unused_parameters
    assert n_bn >= 10 and n_sync == n_bn, (n_bn, n_sync)
    opt = torch.optim.SGD(ddp.parameters(), lr=1e-3, momentum=0.9)
    yy, xx = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")

    def blob(cy, cx, r):
        return (((yy - cy) ** 2 + (xx - cx) ** 2) <= r * r).float()
    targets = []
    for b in range(2):
        cs = [(30 + 5 * rank, 30, 14), (90, 80 - 7 * b, 20), (40, 100, 10)][: 3 - (rank + b) % 2]
        masks = torch.stack([blob(*c) for c in cs])
        centers = torch.tensor([[c[1] / W, c[0] / H] for c in cs], device="cuda").view(len(cs), 1, 2)
        targets.append({"masks": masks, "labels": torch.ones(len(cs), dtype=torch.long, device="cuda"),
                        "fg_masks": (masks.sum(0) > 0).float(), "center_points": centers})
    random.seed(rank)
    ddp.train()
    times = []
    for step in range(args.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses = ddp(vol, targets, True)
        total = sum(v for v in losses.values() if torch.is_tensor(v))
        opt.zero_grad(set_to_none=True)
        total.backward()
        opt.step()
        torch.cuda.synchronize()
        times.append(1e3 * (time.perf_counter() - t0))
    assert torch.isfinite(total)
    grads = {n: p.grad.detach().cpu() for n, p in net.named_parameters() if p.grad is not None}
    stats = {n: b.detach().cpu() for n, b in net.named_buffers() if n.endswith("running_mean") or n.endswith("running_var")}
    params = {n: p.detach().cpu() for n, p in net.named_parameters()}
    torch.save({"grads": grads, "stats": stats, "params": params, "loss": float(total.detach()), "n_sync_bn": n_sync,
                "ms": times, "n_targets": [len(t["labels"]) for t in targets]}, os.path.join(args.out, "rank%d.pt" % rank))
    parallel.shutdown()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--out", default=None, help="directory for the per-rank dumps (default: a temporary directory)")
    ap.add_argument("--graph-decoder", action="store_true",
                    help="capture the decoder core in HIP graphs on every rank (graph.graph_training_decoder) before DDP")
    args = ap.parse_args()
    if "RANK" in os.environ:
        return rank_main(args)
    import tempfile
    if args.out is None:
        args.out = tempfile.mkdtemp(prefix="pct_ddp_")
    os.makedirs(args.out, exist_ok=True)
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    rc = subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.ranks),
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
                          "--ranks", str(args.ranks), "--steps", str(args.steps), "--out", args.out]
                         + (["--graph-decoder"] if args.graph_decoder else []), env=env)
    if rc != 0:
        sys.exit(rc)
    import torch
    r = [torch.load(os.path.join(args.out, "rank%d.pt" % i)) for i in range(args.ranks)]
    worst = 0.0
    for k in r[0]["grads"]:
        for o in r[1:]:
            worst = max(worst, float((r[0]["grads"][k] - o["grads"][k]).abs().max()))
    same_stats = all(torch.equal(r[0]["stats"][k], o["stats"][k]) for k in r[0]["stats"] for o in r[1:])
    same_params = all(torch.equal(r[0]["params"][k], o["params"][k]) for k in r[0]["params"] for o in r[1:])
    summary = {"ranks": args.ranks, "graphed_decoder": bool(args.graph_decoder), "backend": "gloo (ranks share cuda:0)", "sync_batchnorm_modules": r[0]["n_sync_bn"],
               "gradient_tensors": len(r[0]["grads"]), "max_gradient_difference_between_ranks": worst,
               "running_stats_identical": same_stats, "parameters_identical_after_step": same_params,
               "losses": [x["loss"] for x in r], "targets_per_rank": [x["n_targets"] for x in r],
               "ms_per_step_per_rank": [x["ms"] for x in r]}
    print(json.dumps(summary))
    assert worst == 0.0 and same_stats and same_params, summary


if __name__ == "__main__":
    main()
