"""Row a17 (DP gradient mean) on CPU: world_size-2 gloo ranks, flat gradient arena, bucketed all-reduce launched from
post-accumulate-grad hooks (overlap) - the averaged gradients equal the single-process gradient of the full batch."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp_

from mindpose_amd.utils.grad_allreduce import GradientAverager


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(),
                               torch.nn.Linear(64, 8))


def _worker(rank, world, port, out_dir, overlap, mean="pass"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _model()
    avg = GradientAverager(net.parameters(), bucket_mb=0.01, overlap=overlap, mean=mean)  # tiny buckets -> several of them
    assert len(avg.buckets) >= 3
    assert avg.mean_scale == (0.5 if mean == "consumer" else 1.0)
    g = torch.Generator().manual_seed(5)
    x, y = torch.randn(8, 16, generator=g), torch.randn(8, 8, generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    for _ in range(2):  # two steps: the arena is re-armed and zeroed correctly
        avg.begin_step()
        loss = ((net(xs) - ys) ** 2).mean()
        loss.backward()
        avg.finish()
    if rank == 0:
        # what the consumer sees: the arena times the factor it still has to fold in
        torch.save([p.grad.clone() * avg.mean_scale for p in net.parameters()], os.path.join(out_dir, f"grads_{int(overlap)}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_overlapped_allreduce_matches_full_batch(tmp_path):
    net = _model()
    g = torch.Generator().manual_seed(5)
    x, y = torch.randn(8, 16, generator=g), torch.randn(8, 8, generator=g)
    ((net(x) - y) ** 2).mean().backward()
    ref = [p.grad.clone() for p in net.parameters()]
    for overlap in (True, False):
        mp_.spawn(_worker, args=(2, _free_port(), str(tmp_path), overlap), nprocs=2, join=True)
        got = torch.load(os.path.join(tmp_path, f"grads_{int(overlap)}.pt"))
        for a, b in zip(got, ref):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-7)
    # mean="consumer": the collective sums and the optimizer folds 1/world into its gradient scale
    mp_.spawn(_worker, args=(2, _free_port(), str(tmp_path), True, "consumer"), nprocs=2, join=True)
    for a, b in zip(torch.load(os.path.join(tmp_path, "grads_1.pt")), ref):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-7)


def test_mean_modes_are_validated():
    import pytest
    net = _model()
    with pytest.raises(ValueError):
        GradientAverager(net.parameters(), mean="collective")  # no process group: nothing offers an averaging all-reduce
    with pytest.raises(ValueError):
        GradientAverager(net.parameters(), mean="bogus")
    assert GradientAverager(net.parameters(), mean="consumer").mean_scale == 1.0  # world 1


def test_single_process_arena_views():
    net = _model()
    avg = GradientAverager(net.parameters(), bucket_mb=0.01)
    x = torch.randn(4, 16)
    net(x).sum().backward()
    avg.finish()
    total = sum(p.numel() for p in net.parameters())
    assert avg.arena.numel() == total
    for p in net.parameters():
        assert p.grad.data_ptr() >= avg.arena.data_ptr() and p.grad.abs().sum() > 0
    assert abs(float(avg.arena.abs().sum()) - sum(float(p.grad.abs().sum()) for p in net.parameters())) < 1e-3


def _segmented_worker(rank, world, port, out_dir):
    """The segmented step's exchange logic (utils/graph_step.py: plan_bucket_schedule + run_segments) on CPU: one backward pass cut
    into two autograd graphs at a detached leaf, the finished buckets handed to gloo between the segments."""
    from mindpose_amd.utils.graph_step import plan_bucket_schedule, run_segments
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _model()
    avg = GradientAverager(net.parameters(), bucket_mb=0.01, overlap=False)
    assert len(avg.buckets) >= 3 and avg.active
    g = torch.Generator().manual_seed(5)
    x, y = torch.randn(8, 16, generator=g), torch.randn(8, 8, generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    front, back = net[:2], net[2:]  # backward order: `back` (last layers) completes first
    seg_params = [list(back.parameters()), list(front.parameters())]
    schedule = plan_bucket_schedule(avg, seg_params)
    assert sorted(b for r in schedule for b in r) == list(range(len(avg.buckets))) and len(schedule[0]) >= 1
    state = {}

    def seg0():
        avg.arena.zero_()
        h = front(xs)
        leaf = h.detach().requires_grad_()
        loss = ((back(leaf) - ys) ** 2).mean()
        got = torch.autograd.grad([loss], [leaf] + seg_params[0])
        for p, gr in zip(seg_params[0], got[1:]):
            p.grad.add_(gr)
        state["root"], state["seed"] = h, got[0]

    def seg1():
        got = torch.autograd.grad([state["root"]], seg_params[1], grad_outputs=[state["seed"]])
        for p, gr in zip(seg_params[1], got):
            p.grad.add_(gr)

    for _ in range(2):
        issued = run_segments(avg, [seg0, seg1], schedule)
        assert issued >= 0.0 and any(b.get("launched") for b in avg.buckets)
        avg.finish()
    # the same pass without the early hand-over
    segmented = [p.grad.clone() for p in net.parameters()]
    run_segments(avg, [seg0, seg1], schedule, exchange=False)
    assert not any(b.get("launched") for b in avg.buckets)
    avg.finish()
    for a, b in zip(segmented, [p.grad for p in net.parameters()]):
        assert torch.equal(a, b)
    if rank == 0:
        torch.save(segmented, os.path.join(out_dir, "grads_segmented.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_segmented_backward_hands_buckets_over_between_segments_world_size_2_gloo(tmp_path):
    net = _model()
    g = torch.Generator().manual_seed(5)
    x, y = torch.randn(8, 16, generator=g), torch.randn(8, 8, generator=g)
    ((net(x) - y) ** 2).mean().backward()
    ref = [p.grad.clone() for p in net.parameters()]
    mp_.spawn(_segmented_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = torch.load(os.path.join(tmp_path, "grads_segmented.pt"))
    for a, b in zip(got, ref):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-7)
