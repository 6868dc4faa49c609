"""CPU, world_size 2, gloo: the multi-GPU launch semantics of the path -- one process per device,
images sharded across ranks, DDP(find_unused_parameters=True) as the reference's make_parallel does (connectomics/model/build.py:74-102), env:// rendezvous as connectomics/utils/system.py:58-70.
Forward needs no collective; backward all-reduces gradients: after one step both ranks must hold identical gradients
equal to the mean of the per-shard gradients computed without DDP."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build():
    from pctrans_amd.config import get_cfg, resnet_output_shape
    from pctrans_amd.meta_arch.mask_former_head import MaskFormerHead
    torch.manual_seed(0)
    cfg = get_cfg(num_queries=6, norm="BN", sem_norm="BN", enc_layers=2, dec_layers=3)
    shapes = resnet_output_shape(18)
    return MaskFormerHead(**MaskFormerHead.from_config(cfg, shapes)), shapes


def _shard(rank, shapes):
    g = torch.Generator().manual_seed(100 + rank)
    return {k: torch.randn(1, s.channels, 64 // s.stride, 64 // s.stride, generator=g) for k, s in shapes.items()}


def _loss(pred):
    return pred["pred_masks"].square().mean() + pred["reference_points"].sum() + pred["sem_mask"].mean()


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from pctrans_amd.pixel_decoder.ops.modules import ms_deform_attn as msda_mod
    msda_mod.allow_cpu_reference(True)
    dist.init_process_group("gloo", init_method="env://")
    head, shapes = _build()
    head = head.eval()      # SyncBatchNorm conversion (build.py:80-81) is GPU-only in torch; BN in eval = same math
    ddp = torch.nn.parallel.DistributedDataParallel(head, find_unused_parameters=True)
    pred, _ = ddp(_shard(rank, shapes))
    _loss(pred).backward()
    grads = {n: p.grad.clone() for n, p in head.named_parameters() if p.grad is not None}
    torch.save(grads, os.path.join(out_dir, "g%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_ddp_two_ranks_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g0 = torch.load(tmp_path / "g0.pt")
    g1 = torch.load(tmp_path / "g1.pt")
    assert g0.keys() == g1.keys() and len(g0) > 100
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k              # all-reduced: bitwise identical on both ranks

    # same thing without DDP: mean of the two per-shard gradients
    from pctrans_amd.pixel_decoder.ops.modules import ms_deform_attn as msda_mod
    prev = msda_mod.allow_cpu_reference(True)
    try:
        want = None
        for r in range(world):
            head, shapes = _build()
            head.eval()
            pred, _ = head(_shard(r, shapes))
            _loss(pred).backward()
            g = {n: p.grad for n, p in head.named_parameters() if p.grad is not None}
            want = g if want is None else {k: want[k] + g[k] for k in g}
        for k in g0:
            torch.testing.assert_close(g0[k], want[k] / world, rtol=1e-4, atol=1e-6, msg=k)
    finally:
        msda_mod.allow_cpu_reference(prev)
