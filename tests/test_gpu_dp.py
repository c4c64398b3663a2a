"""Row a17 on the GPU: the data-parallel step tail through the C ABI (mp_grad_finite_check, mp_adamw_step_scaled, the RCCL
entries mp_comm_* / mp_allreduce_grads) and the gradient mean of an HRNet-W32 training step over RCCL.

One-GPU box: the RCCL path runs on a ONE-rank group (same calls, same streams, nothing to exchange).  Two or more GPUs (skipped
otherwise): two "nccl" ranks, each on its own device, eager-overlapped and graph-captured steps, torch and native transports -
the reduced arena equals the mean of the ranks' local arenas (per-device BatchNorm statistics as in the reference, so the
yardstick is the mean of the LOCAL gradients, not a full-batch gradient).  Reference: tools/train.py:43-49."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_finite_check_and_scaled_adamw_vs_torch():
    from mindpose_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    n = 100_003  # not a multiple of 4: the scalar tail of the check is exercised
    grad = torch.randn(n, generator=g).to(dev) * 4096.0
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.mp_grad_finite_check(grad.data_ptr(), n, flag.data_ptr(), _lib.stream()), "finite")
    assert int(flag.item()) == 0
    for pos, bad in ((5, float("inf")), (n - 1, float("nan")), (n // 2, float("-inf"))):
        gb = grad.clone()
        gb[pos] = bad
        flag.zero_()
        _lib.check(lib.mp_grad_finite_check(gb.data_ptr(), n, flag.data_ptr(), _lib.stream()), "finite")
        assert int(flag.item()) == 1, (pos, bad)
    p = torch.randn(n, generator=g).to(dev)
    m = torch.randn(n, generator=g).to(dev) * 0.1
    v = torch.rand(n, generator=g).to(dev) * 0.1
    lr, b1, b2, eps, wd, scale = 1e-3, 0.9, 0.999, 1e-6, 0.05, 1.0 / (4096.0 * 2)
    # reference: the unscaled kernel on a pre-scaled gradient (what the optimizer did before the fold)
    p0, m0, v0 = p.clone(), m.clone(), v.clone()
    _lib.check(lib.mp_adamw_step(p0.data_ptr(), (grad * scale).data_ptr(), m0.data_ptr(), v0.data_ptr(), n, lr, b1, b2, eps, wd,
                                 _lib.stream()), "adamw")
    p1, m1, v1 = p.clone(), m.clone(), v.clone()
    _lib.check(lib.mp_adamw_step_scaled(p1.data_ptr(), grad.data_ptr(), m1.data_ptr(), v1.data_ptr(), n, lr, b1, b2, eps, wd, scale,
                                        None, _lib.stream()), "adamw scaled")
    assert torch.equal(p0, p1) and torch.equal(m0, m1) and torch.equal(v0, v1)  # power-of-two scale: bit-identical
    # and against the formula in torch (mindspore.nn.AdamWeightDecay: no bias correction)
    gs = grad * scale
    mt = b1 * m + (1 - b1) * gs
    vt = b2 * v + (1 - b2) * gs * gs
    pt = p - lr * (mt / (vt.sqrt() + eps) + wd * p)
    assert torch.allclose(p1, pt, rtol=1e-6, atol=1e-7)
    # skip flag set: nothing moves
    flag.fill_(1)
    p2, m2, v2 = p.clone(), m.clone(), v.clone()
    _lib.check(lib.mp_adamw_step_scaled(p2.data_ptr(), grad.data_ptr(), m2.data_ptr(), v2.data_ptr(), n, lr, b1, b2, eps, wd, scale,
                                        flag.data_ptr(), _lib.stream()), "adamw skip")
    assert torch.equal(p2, p) and torch.equal(m2, m) and torch.equal(v2, v)


def test_native_comm_one_rank():
    """mp_comm_get_unique_id / mp_comm_init_rank / mp_allreduce_grads / mp_reduce_scatter_allgather_grads on a one-rank
    communicator: sum and mean leave the buffer unchanged; the calls are asynchronous on the communicator's stream."""
    from mindpose_amd import _lib
    from mindpose_amd.utils.grad_allreduce import NativeComm
    lib = _lib.load()
    assert lib.mp_comm_available() == 1
    dev = torch.device("cuda:0")
    comm = NativeComm(dev)
    x = torch.randn(1 << 20, device=dev)
    ref = x.clone()
    for average in (False, True):
        for split in (False, True):
            comm.all_reduce(x, average=average, split=split).synchronize()
            assert torch.equal(x, ref)
    comm.close()


def _build(dev, amp, overlap, transport, force=False):
    import mindpose_amd as mp
    from mindpose_amd.utils import AdamWeightDecay
    torch.manual_seed(0)
    net = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=0).to(dev).train()
    if amp:
        mp.models.auto_mixed_precision(net, "O2")
    nwl = mp.create_network_with_loss(net, mp.create_loss("joint_mse", use_target_weight=True), has_extra_inputs=True)
    opt = AdamWeightDecay(net, lr=1e-3, weight_decay=0.05, filter_bias_and_bn=True, overlap=overlap, transport=transport,
                          bucket_mb=8.0, force_collectives=force)
    return mp, net, nwl, opt


def _batch(mp, dev, rank, n=4, hw=64):
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(n, 3, hw, hw, generator=g).to(dev)
    kp = (torch.rand(n, 17, 3, generator=g) * torch.tensor([float(hw), float(hw), 2.0])).to(dev)
    tgt = mp.TopDownGenerateTarget(config=dict(image_size=[hw, hw], heatmap_size=[hw // 4, hw // 4]), sigma=2.0)
    return (x, *tgt(kp))


def _dp_worker(rank, world, port, out_dir):
    """Every DP flavour on this rank's own GPU; results -> out_dir/rank<r>.pt (asserts inside raise in the parent through spawn)."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", rank if torch.cuda.device_count() >= world else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from mindpose_amd.utils import DynamicLossScaleManager, GraphedTrainStep
    report = {}
    for name, amp, overlap, transport, graphed in (("eager_overlap_torch", False, True, "torch", False),
                                                   ("eager_overlap_native", False, True, "native", False),
                                                   ("graph_torch_o2", True, False, "torch", True),
                                                   ("graph_native_o2", True, False, "native", True)):
        mp, net, nwl, opt = _build(dev, amp, overlap, transport, force=world == 1)
        x, target, weight = _batch(mp, dev, rank)
        scaler = DynamicLossScaleManager(init_loss_scale=1024.0) if amp else None
        if graphed:
            step = GraphedTrainStep(nwl, opt, (x, target, weight), loss_scale_manager=scaler, warmup=2)
            # local arena of this rank: one replay without the exchange
            step.replay(exchange=False)
            torch.cuda.synchronize()
            local = opt.grads.arena.clone()
            # under an active exchange the step is captured in segments: buckets leave for the all-reduce between them
            assert step.segments == 4 and len(step.graphs) == 4
            early = sum(len(r) for r in step.bucket_schedule[:-1])
            assert sorted(b for r in step.bucket_schedule for b in r) == list(range(len(opt.grads.buckets)))
            assert early >= len(opt.grads.buckets) // 2, step.bucket_schedule
            report.setdefault("bucket_schedule", {})[name] = [len(r) for r in step.bucket_schedule]
            before = opt.flat.clone()
            loss = step(x, target, weight)
            assert step.updated
        else:
            opt.zero_grad()
            loss = nwl(x, target, weight)
            # local gradients: a second, hook-free backward pass is not available - compute them with the exchange disabled
            hooks, opt.grads._hooks = opt.grads._hooks, []
            for h in hooks:
                h.remove()
            loss.backward()
            local = opt.grads.arena.clone()
            # now the real exchange, non-overlapped launch of every bucket from finish()
            opt.grads.overlap = False
            before = opt.flat.clone()
            opt.step()
        torch.cuda.synchronize()
        reduced = opt.grads.arena.clone() * opt.grads.mean_scale
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        mean_local = torch.stack(gathered).double().mean(0).float()
        err = float((reduced - mean_local).abs().max() / mean_local.abs().max())
        assert err < 1e-6, (name, err)
        assert not torch.equal(opt.flat, before), name
        # every rank holds the same parameters after the update
        flats = [torch.empty_like(opt.flat) for _ in range(world)]
        dist.all_gather(flats, opt.flat)
        assert all(torch.equal(flats[0], f) for f in flats[1:]), name
        report[name] = dict(err=err, loss=float(loss.detach()))
        if opt.grads.native is not None:
            opt.grads.native.close()
        del loss
        if graphed:
            del step
    # the segmented capture is the single backward pass cut in pieces: its gradient arena equals the one-graph step's BIT FOR BIT
    arenas = {}
    for segs in (1, 2, 3, 4):  # 2 / 3: groups merged into one segment stay one autograd graph (only the boundaries BETWEEN segments are cut)
        mp, net, nwl, opt = _build(dev, True, False, "torch", force=world == 1)
        x, target, weight = _batch(mp, dev, rank, n=6, hw=128)
        step = GraphedTrainStep(nwl, opt, (x, target, weight), loss_scale_manager=DynamicLossScaleManager(init_loss_scale=1024.0),
                                warmup=2, segments=segs)
        assert step.segments == segs
        step.replay(exchange=False)
        torch.cuda.synchronize()
        arenas[segs] = (opt.grads.arena.clone(), float(step.static_loss))
        if segs == 4:
            step(x, target, weight)  # a full step: buckets launched between the segments, the rest from finish()
            assert step.updated and len(step.issue_ms) == 2
        opt.close()
        del step
    for segs in (2, 3, 4):
        assert arenas[1][1] == arenas[segs][1]
        assert torch.equal(arenas[1][0], arenas[segs][0]), f"{segs}-segment backward differs from the one-graph backward"
    report["segmented_equals_single"] = True
    # eager step WITH the overlap hooks live (bucket all-reduces launched from backward): equals the non-overlapped result
    mp, net, nwl, opt = _build(dev, False, True, "torch", force=world == 1)
    x, target, weight = _batch(mp, dev, rank)
    opt.zero_grad()
    nwl(x, target, weight).backward()
    opt.grads.finish()
    overlapped = opt.grads.arena.clone()
    mp, net, nwl, opt2 = _build(dev, False, False, "torch", force=world == 1)
    opt2.zero_grad()
    nwl(x, target, weight).backward()
    opt2.grads.finish()
    assert torch.equal(overlapped, opt2.grads.arena), "overlapped bucket all-reduce differs from the single launch"
    report["overlap_equals_single"] = True
    torch.save(report, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _run_dp(world, tmp_path):
    import torch.multiprocessing as mp_
    mp_.spawn(_dp_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    reports = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    assert all(rep["overlap_equals_single"] and rep["segmented_equals_single"] for rep in reports)
    return reports


def test_one_rank_nccl_group_runs_the_collective_path(tmp_path):
    """The whole DP step tail on a ONE-rank RCCL group (collectives forced): what a one-GPU box can run of config 4's exchange."""
    reports = _run_dp(1, tmp_path)
    assert set(reports[0]) >= {"eager_overlap_torch", "eager_overlap_native", "graph_torch_o2", "graph_native_o2"}


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (one rank per device over RCCL/xGMI)")
def test_two_rank_nccl_hrnet_w32_step(tmp_path):
    reports = _run_dp(2, tmp_path)
    # different data per rank: the losses differ, the reduced arenas agree with the mean of the local ones (checked in-worker)
    assert reports[0]["graph_torch_o2"]["loss"] != reports[1]["graph_torch_o2"]["loss"]


def test_bench_two_rank_rehearsal_on_one_gpu_reports_the_dp_leg():
    """`bench.py --gpus 2` as the driver launches it (bench.py spawns its ranks through torch.distributed.run), rehearsed on ONE GPU:
    MINDPOSE_BENCH_SHARED_GPU_REHEARSAL=1 puts both ranks on device 0 with gloo for the barrier / MAX / gradient exchange.  Not a
    measurement - the first real multi-GPU run must not fail on plumbing: the line parses, names two ranks and carries the
    data-parallel training leg with its exchange figures (segmented backward, buckets released between the segments)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MINDPOSE_BENCH_SHARED_GPU_REHEARSAL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # a child process tree of its own: nothing here has to be exec'ed from this (GPU-initialised) process
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-roofline",
                           "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["global_batch"] == 2 * line["config"]["per_gpu_batch"]
    leg = line["extra_workloads"]["config3_train_ampO2_dp"]
    assert "error" not in leg, leg
    for key in ("rccl_nranks", "transport", "allreduce_ms_per_step", "allreduce_issue_ms", "backward_segments", "buckets_released_per_segment",
                "gradient_bytes", "buckets", "value", "ms_per_step"):
        assert key in leg, key
    assert leg["rccl_nranks"] == 2 and leg["global_batch"] == 2 * leg["per_gpu_batch"]
    assert leg["backward_segments"] == 4 and len(leg["buckets_released_per_segment"]) == 4
    assert sum(leg["buckets_released_per_segment"]) == leg["buckets"]  # every bucket leaves exactly once
    assert leg["allreduce_ms_per_step"] is not None and leg["value"] > 0
