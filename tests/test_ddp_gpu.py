"""GPU, two ranks sharing the one card over gloo: a TRAINING step of the whole model through parallel.make_parallel with
norm_mode="sync_bn" (connectomics/model/build.py:74-102; configs/CVPPP/CVPPP-PCTrans.yaml:15,24 `NORM: SyncBN`), i.e. the
SyncBatchNorm statistics exchange, DDP's gradient buckets and the criterion's num_masks all-reduce on device tensors.
(The multi-GPU curve itself is the driver's to measure; this executes the code path.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_ranks_train_one_step_with_syncbn_and_ddp_on_device_tensors(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rehearse_ddp_sync_bn.py"), "--ranks", "2", "--steps", "2",
                          "--out", str(tmp_path)], capture_output=True, text=True, timeout=850)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    s = json.loads(out.stdout.strip().splitlines()[-1])
    assert s["ranks"] == 2 and s["sync_batchnorm_modules"] >= 10 and s["gradient_tensors"] > 100
    assert s["max_gradient_difference_between_ranks"] == 0.0          # all-reduced: bitwise identical on both ranks
    assert s["running_stats_identical"] and s["parameters_identical_after_step"]
    assert s["targets_per_rank"][0] != s["targets_per_rank"][1]       # the shards really differ (num_masks is all-reduced)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_ranks_train_with_the_decoder_core_replayed_from_hip_graphs(tmp_path):
    """graph.graph_training_decoder under the reference's launch: SyncBatchNorm conversion, capture of the (norm-free) decoder
    core on every rank, then DDP -- three optimiser steps, gradients / statistics / parameters bitwise equal across the ranks
    (DDP's bucket hooks fire on the gradients the graphs return), losses finite."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rehearse_ddp_sync_bn.py"), "--ranks", "2", "--steps", "3",
                          "--graph-decoder", "--out", str(tmp_path)], capture_output=True, text=True, timeout=850)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    s = json.loads(out.stdout.strip().splitlines()[-1])
    assert s["ranks"] == 2 and s["graphed_decoder"] and s["sync_batchnorm_modules"] >= 10 and s["gradient_tensors"] > 100
    assert s["max_gradient_difference_between_ranks"] == 0.0
    assert s["running_stats_identical"] and s["parameters_identical_after_step"]
    assert all(l == l and abs(l) < 1e6 for l in s["losses"])
