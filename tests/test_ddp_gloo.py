"""CPU, world_size 2, gloo: the multi-GPU launch semantics of the path -- one process per device,
images sharded across ranks, through pctrans_amd.parallel: init_devices (env:// rendezvous, connectomics/utils/system.py:58-70)
and make_parallel (SyncBatchNorm conversion + DDP(find_unused_parameters=True), connectomics/model/build.py:74-102).
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
    from pctrans_amd import parallel
    device, r, lr, w = parallel.init_devices(backend="gloo")
    assert (device.type, r, lr, w) == ("cpu", rank, rank, world) and dist.get_backend() == "gloo"
    head, shapes = _build()
    ddp = parallel.make_parallel(head, device, parallel="DDP", norm_mode="sync_bn")
    assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel) and ddp.find_unused_parameters
    # the shipped configs' BatchNorms (NORM / SEMANTIC_NORM: SyncBN) become SyncBatchNorm as build.py:80-81 does -- on
    # GPU ranks; torch's DDP refuses SyncBatchNorm on CPU modules, so this rehearsal keeps BatchNorm (same math in eval)
    nbn = sum(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) for m in head.modules())
    assert nbn >= 2 and sum(isinstance(m, torch.nn.SyncBatchNorm) for m in parallel.convert_norms(head).modules()) == nbn
    ddp.eval()
    head = ddp.module
    pred, _ = ddp(_shard(rank, shapes))
    _loss(pred).backward()
    grads = {n: p.grad.clone() for n, p in head.named_parameters() if p.grad is not None}
    torch.save(grads, os.path.join(out_dir, "g%d.pt" % rank))
    parallel.shutdown()


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


def _run_bench(extra_env, *argv):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), env=env, capture_output=True,
                          text=True, timeout=600)


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-box behaviour")
@pytest.mark.timeout(900)
def test_bench_gpus_n_never_reports_one_rank_as_n():
    """`python bench.py --gpus 2` from a bare shell must either run 2 ranks or fail: never rc 0 with n_gpus 1."""
    r = _run_bench({}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and '"n_gpus"' not in r.stdout, (r.returncode, r.stdout[-300:])
    assert "GPU(s) visible" in r.stderr
    # with the rehearsal switch the parent does start torch.distributed.run; on a box without a GPU both ranks fail
    # their "needs an MI355X" check and the parent reports the children's failure
    r = _run_bench({"PCT_BENCH_SHARE_GPU": "1"}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and '"n_gpus"' not in r.stdout, (r.returncode, r.stdout[-300:])
    assert "needs an MI355X" in r.stderr
    # a launcher/--gpus mismatch is an error too
    r = _run_bench({"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "2")
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


def test_convert_norms_refuses_a_model_whose_training_front_is_captured():
    """graph_training_front captures the per-rank BatchNorm kernels; converting to SyncBatchNorm afterwards
    (make_parallel's default, build.py:80-81) would leave them unsynchronised -- convert_norms must refuse."""
    from pctrans_amd import parallel
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), torch.nn.BatchNorm2d(4))
    assert not parallel.has_graphed_front(m)
    assert isinstance(parallel.convert_norms(m)[1], torch.nn.SyncBatchNorm)
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), torch.nn.BatchNorm2d(4))
    m[0].__dict__["_pct_graphed"] = (None, None)                 # what graph_training_front leaves on a captured module
    assert parallel.has_graphed_front(m)
    with pytest.raises(RuntimeError, match="captured in HIP graphs"):
        parallel.convert_norms(m)
    frozen = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), torch.nn.GroupNorm(2, 4))
    frozen[0].__dict__["_pct_graphed"] = (None, None)
    parallel.convert_norms(frozen)                               # nothing to convert: allowed


def test_every_spawned_rank_gets_its_own_miopen_database_and_cache(tmp_path):
    """eight ranks running MIOpen's full find at once must not share one user database / kernel cache.
    bench.py sets per-rank directories at import time (before torch / MIOpen are touched) from LOCAL_RANK."""
    import subprocess
    import sys
    from conftest import ROOT
    seen = {}
    for r in (0, 1, 7):
        env = dict(os.environ, WORLD_SIZE="8", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT="1",
                   PCT_BENCH_MIOPEN_BASE=str(tmp_path))
        env.pop("MIOPEN_USER_DB_PATH", None)
        env.pop("MIOPEN_CUSTOM_CACHE_DIR", None)
        out = subprocess.run([sys.executable, "-c", "import os, bench; print(os.environ['MIOPEN_USER_DB_PATH']); "
                              "print(os.environ['MIOPEN_CUSTOM_CACHE_DIR'])"], cwd=ROOT, env=env, capture_output=True,
                             text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-500:]
        db, cache = out.stdout.strip().splitlines()[-2:]
        assert ("rank%d" % r) in db and ("rank%d" % r) in cache and os.path.isdir(db) and os.path.isdir(cache)
        seen[r] = (db, cache)
    assert len({v[0] for v in seen.values()}) == 3 and len({v[1] for v in seen.values()}) == 3
    # a single process (the driver's N = 1 run) keeps MIOpen's defaults; an operator's own setting is respected
    env = dict(os.environ, PCT_BENCH_MIOPEN_BASE=str(tmp_path))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MIOPEN_USER_DB_PATH", "MIOPEN_CUSTOM_CACHE_DIR"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-c", "import os, bench; print(os.environ.get('MIOPEN_USER_DB_PATH'))"], cwd=ROOT,
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.stdout.strip().splitlines()[-1] == "None"
    env = dict(os.environ, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", MIOPEN_USER_DB_PATH="/somewhere/else")
    out = subprocess.run([sys.executable, "-c", "import os, bench; print(os.environ['MIOPEN_USER_DB_PATH'])"], cwd=ROOT,
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.stdout.strip().splitlines()[-1] == "/somewhere/else"
