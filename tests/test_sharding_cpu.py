"""N>1 path on CPU: world_size-2 gloo ranks shard crops, run their share of the (oracle) decode, gather,
and the union equals the single-process result; bench.py's timing reduction (MAX over ranks) is exercised."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp_

from mindpose_amd.utils import shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import decoder as od
    from tests.golden import recipes
    hm = recipes.uniform_heatmaps(total, 17, 64, 48, 123)
    center, scale, score = recipes.boxes(total, 124)
    b, e = shard_range(total, world, rank)
    preds, boxes, _ = od.decode(hm[b:e], center[b:e], scale[b:e], score[b:e], shift_coord=True)
    gathered = [None] * world
    dist.all_gather_object(gathered, (b, e, preds))
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # bench.py: elapsed = MAX over ranks
    dist.barrier()
    if rank == 0:
        full = np.concatenate([g[2] for g in sorted(gathered, key=lambda g: g[0])])
        np.save(os.path.join(out_dir, "gathered.npy"), full)
        np.save(os.path.join(out_dir, "tmax.npy"), t.numpy())
    dist.destroy_process_group()


def test_shard_range_partitions():
    for total in (0, 1, 7, 128, 129):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_world_size_2_gloo_sharded_decode(tmp_path):
    from oracle import decoder as od
    from tests.golden import recipes
    total, world = 9, 2
    port = _free_port()
    mp_.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    hm = recipes.uniform_heatmaps(total, 17, 64, 48, 123)
    center, scale, score = recipes.boxes(total, 124)
    ref, _, _ = od.decode(hm, center, scale, score, shift_coord=True)
    assert np.array_equal(np.load(tmp_path / "gathered.npy"), ref)
    assert np.load(tmp_path / "tmax.npy")[0] == 0.2


def test_lr_schedules_match_reference_formulas():
    import math
    from mindpose_amd.utils import WarmupCosineDecayLR, WarmupMultiStepDecayLR, create_lr_scheduler
    # reference recipe: lr 1e-3, warm-up 500 steps, milestones [170, 200] of 210 epochs (configs/hrnet/hrnet_w32_ascend.yaml:75-88)
    sched = create_lr_scheduler("warmup_multi_step_decay", lr=1e-3, total_epochs=210, steps_per_epoch=100, warmup=500,
                                milestones=[170, 200], decay_rate=0.1)
    assert isinstance(sched, WarmupMultiStepDecayLR)
    assert sched(0) == 0.0 and abs(sched(250) - 5e-4) < 1e-12 and sched(500) == 1e-3 and sched(501) == 1e-3
    assert sched(169 * 100 - 1) == 1e-3 and sched(169 * 100) == 1e-3 * 0.1
    assert sched(199 * 100 - 1) == 1e-3 * 0.1 and sched(199 * 100) == 1e-3 * 0.1 * 0.1 and sched(210 * 100 - 1) == 1e-3 * 0.1 * 0.1
    # fractional warm-up = fraction of the total steps; cosine decay over the rest, never below min_lr
    cos = create_lr_scheduler("warmup_cosine_decay", lr=2e-3, total_epochs=10, steps_per_epoch=100, warmup=0.1, min_lr=1e-5)
    assert isinstance(cos, WarmupCosineDecayLR) and cos.warmup_steps == 100 and cos.decay_steps == 900
    assert abs(cos(50) - 1e-3) < 1e-15 and cos(100) == 2e-3
    want = 1e-5 + 0.5 * (2e-3 - 1e-5) * (1 + math.cos(math.pi * 450 / 900))
    assert abs(cos(550) - want) < 1e-15 and abs(cos(1000) - 1e-5) < 1e-12
    import pytest
    with pytest.raises(ValueError, match="Warmup steps"):
        WarmupMultiStepDecayLR(1e-3, 1, 10, [1], warmup=11)


def test_dynamic_loss_scale_manager_semantics():
    from mindpose_amd.utils import DynamicLossScaleManager
    m = DynamicLossScaleManager(init_loss_scale=2.0 ** 10, scale_factor=2.0, scale_window=3)
    assert m.get_loss_scale() == 1024.0 and m.scale(0.5) == 512.0
    m.update_loss_scale(True)                       # overflow: halve, remember the iteration
    assert m.loss_scale == 512.0 and m.skipped_steps == 1
    for _ in range(2):
        m.update_loss_scale(False)
    assert m.loss_scale == 512.0                    # window not reached yet
    m.update_loss_scale(False)                      # 3 clean steps since the overflow: double
    assert m.loss_scale == 1024.0
    for _ in range(40):                             # never below 1
        m.update_loss_scale(True)
    assert m.loss_scale == 1.0
    import pytest
    with pytest.raises(ValueError):
        DynamicLossScaleManager(init_loss_scale=0.5)


def test_bench_spawns_its_own_ranks_without_a_launcher(monkeypatch):
    """`python bench.py --gpus N` with no WORLD_SIZE starts the ranks itself through torch.distributed.run (before any GPU call);
    with WORLD_SIZE set it is a rank and must not spawn."""
    import importlib
    import sys
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    import pytest
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 7  # the children's status is the parent's
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def _dp_leg_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib
    import json
    from mindpose_amd.utils.grad_allreduce import GradientAverager
    bench = importlib.import_module("bench")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))

    class Opt:  # the slice of the arena optimizers dp_leg_report reads
        def __init__(self):
            self.grads = GradientAverager(model.parameters(), bucket_mb=1e-4, overlap=False, mean="pass")
            self.time_comm, self.comm_events = False, []

    opt = Opt()
    x = torch.randn(5, 8, generator=torch.Generator().manual_seed(rank))

    def step():
        opt.grads.begin_step()
        model(x).square().mean().backward()
        opt.grads.finish()

    rep = bench.dp_leg_report(step, opt, dist, world, rank, torch.device("cpu"), per_gpu_batch=5, steps=3, warmup=1)
    # after the leg every rank holds the MEAN gradient
    g = opt.grads.arena.clone()
    ref = [torch.zeros_like(g) for _ in range(world)]
    dist.all_gather(ref, g)
    assert all(torch.equal(r, g) for r in ref)
    if rank == 0:
        with open(os.path.join(out_dir, "dp_leg.json"), "w") as f:
            json.dump(rep, f)
    dist.destroy_process_group()


def test_dp_training_leg_report_keys_world_size_2_gloo(tmp_path):
    """bench.py attaches `extra_workloads.config3_train_ampO2_dp` on multi-rank runs: the reporting code on two
    gloo ranks with a stand-in model - keys, the communicator's own rank count, aggregate throughput."""
    import json
    port = _free_port()
    mp_.spawn(_dp_leg_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    rep = json.load(open(tmp_path / "dp_leg.json"))
    for key in ("value", "unit", "ms_per_step", "steps", "warmup", "per_gpu_batch", "global_batch", "allreduce_ms_per_step", "rccl_nranks",
                "transport", "gradient_bytes", "buckets", "mean", "n_gpus_seen"):
        assert key in rep, key
    assert rep["rccl_nranks"] == 2 and rep["global_batch"] == 10 and rep["steps"] == 3 and "gloo" in rep["transport"]
    assert rep["value"] > 0 and rep["buckets"] >= 2 and rep["gradient_bytes"] == 4 * (8 * 16 + 16 + 16 * 4 + 4)
    assert rep["allreduce_ms_per_step"] is None  # no device events on CPU tensors
