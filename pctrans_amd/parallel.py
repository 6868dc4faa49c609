"""Process-group set-up and the data-parallel wrapper of the path: one process per GPU, images sharded across ranks.

Host-side mirror of the reference's launch semantics:
  * `init_devices`  -- connectomics/utils/system.py:53-95: `dist.init_process_group(backend, 'env://')` from the
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* variables a `torch.distributed.run` launch exports
    (README.md:30-33 uses the older `torch.distributed.launch --nproc_per_node=4 ... --distributed`), one device
    per process (`set_device(local_rank)`), seed = local rank unless given.
  * `make_parallel` -- connectomics/model/build.py:74-102: BatchNorm -> SyncBatchNorm when the norm mode asks for it
    (configs/CVPPP/CVPPP-PCTrans.yaml:15,24 `NORM: SyncBN`), then `DistributedDataParallel(device_ids=[local_rank],
    find_unused_parameters=True)` (build.py:88 forces True).

Backend string 'nccl' is RCCL on PyTorch-ROCm (xGMI inside a node); 'gloo' is the CPU rehearsal the tests use.
The forward path has no data-path collective: the only collectives are DDP's bucketed gradient all-reduce, the
SyncBatchNorm statistics in training and the criterion's `num_masks` all-reduce.
"""
import os
import random

import numpy as np
import torch
import torch.distributed as dist
from torch import nn

_ENV_KEYS = ("MASTER_ADDR", "MASTER_PORT", "RANK", "LOCAL_RANK", "WORLD_SIZE")


def init_seed(seed):
    """system.py:46-50."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def init_devices(distributed=None, backend="nccl", manual_seed=None, share_gpu=False):
    """Returns (device, rank, local_rank, world_size).

    distributed=None follows the environment: a process started by `torch.distributed.run` (WORLD_SIZE > 1) joins the
    group, anything else is a single process.  With backend 'nccl' every rank owns GPU `local_rank`; `share_gpu`
    (rehearsal on a 1-GPU box) folds the local rank onto the devices that exist.  backend 'gloo' without CUDA gives CPU
    devices (tests)."""
    if distributed is None:
        distributed = int(os.environ.get("WORLD_SIZE", "1")) > 1
    if not distributed:
        seed = 0 if manual_seed is None else manual_seed
        device = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
        init_seed(seed)
        return device, 0, 0, 1
    missing = [k for k in _ENV_KEYS if k not in os.environ]
    if missing:
        raise RuntimeError("distributed launch needs %s in the environment (start with `python -m "
                           "torch.distributed.run --nproc-per-node N ...`)" % ", ".join(missing))
    rank, local_rank = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])
    if backend == "nccl":
        assert torch.cuda.is_available(), "Distributed training without GPUs is not supported!"   # system.py:55-56
    use_cuda = torch.cuda.is_available() and (backend == "nccl" or share_gpu)
    if use_cuda:
        ndev = torch.cuda.device_count()
        if share_gpu:
            local_rank_dev = local_rank % ndev
        else:
            if local_rank >= ndev:
                raise RuntimeError("LOCAL_RANK %d but only %d GPU(s) visible" % (local_rank, ndev))
            local_rank_dev = local_rank
        torch.cuda.set_device(local_rank_dev)
        device = torch.device("cuda", local_rank_dev)
    else:
        device = torch.device("cpu")
    if not dist.is_initialized():
        if backend == "nccl" and not share_gpu:
            dist.init_process_group("nccl", init_method="env://", device_id=device)
        else:
            dist.init_process_group("gloo" if share_gpu else backend, init_method="env://")
    assert dist.get_world_size() == int(os.environ["WORLD_SIZE"]) and dist.get_rank() == rank
    init_seed(local_rank if manual_seed is None else manual_seed)
    return device, rank, local_rank, dist.get_world_size()


def has_graphed_front(model):
    """True when pctrans_amd.graph.graph_training_front / graph_training_decoder captured part of this model into HIP graphs."""
    return any("_pct_graphed" in m.__dict__ or "_pct_graphed_core" in m.__dict__ for m in model.modules())


def convert_norms(model):
    """build.py:80-81: every BatchNorm*d of the model becomes a SyncBatchNorm (the backbone's FrozenBN is not a
    BatchNorm and stays as it is).  Refused for a model whose training front is captured: the graphs would go on
    replaying the per-rank BatchNorm kernels behind the converted modules."""
    if has_graphed_front(model) and any(isinstance(m, nn.modules.batchnorm._BatchNorm) for m in model.modules()):
        raise RuntimeError("convert_norms: this model's training front is captured in HIP graphs (graph_training_front); "
                           "converting its BatchNorms to SyncBatchNorm now would leave the captured kernels unsynchronised. "
                           "Convert first and capture only norm-free / frozen-norm fronts, or do not capture.")
    return nn.SyncBatchNorm.convert_sync_batchnorm(model)


def make_parallel(model, device, parallel="DDP", norm_mode="sync_bn", find_unused_parameters=True):
    """build.py:74-102.  parallel: 'DDP' | 'DP' | anything else = single device."""
    if parallel == "DDP":
        if not dist.is_initialized():
            raise RuntimeError("make_parallel('DDP') needs an initialised process group (init_devices)")
        # torch's DistributedDataParallel refuses SyncBatchNorm on CPU modules ("only work with GPU modules"): the gloo
        # rehearsal keeps plain BatchNorm (identical in eval; per-rank statistics in training)
        if norm_mode == "sync_bn" and device.type == "cuda":
            model = convert_norms(model)
        model = model.to(device)
        if device.type == "cuda":
            return nn.parallel.DistributedDataParallel(model, device_ids=[device.index], output_device=device.index,
                                                       find_unused_parameters=True)
        return nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)
    if parallel == "DP":
        # build.py:92-96.  Single-process multi-thread replication: the library's diagnostic hooks (kernel choice override,
        # last-kernel report, per-launch timing) are process-wide and are not meant to be flipped while DataParallel's
        # worker threads launch; the kernels themselves keep no per-call host state.  The north-star launch is DDP.
        return nn.DataParallel(model.to(device), device_ids=list(range(torch.cuda.device_count())))
    return model.to(device)


def shutdown():
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
