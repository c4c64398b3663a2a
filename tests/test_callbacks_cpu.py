"""N4 remainder: the training loop's evaluation / checkpoint callback and the loss helpers, against the behaviour of
mindpose/callbacks/eval_callback.py:16-202 and mindpose/utils/misc.py:7-36 - host logic only, CPU."""
import json
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as tmp

from mindpose_amd.callbacks import EvalCallback
from mindpose_amd.utils import load_checkpoint
from mindpose_amd.utils.misc import Allreduce, AverageMeter


def test_average_meter_matches_the_reference_recurrence():
    m = AverageMeter()
    vals = [0.5, 0.25, 1.0, 0.125]
    for v in vals:
        m.update(torch.tensor(v))
    assert float(m.val) == vals[-1] and m.count == 4 and abs(float(m.avg) - np.mean(vals)) < 1e-12
    m.update(torch.tensor([1.0, 3.0]), n=2)  # a loss with two items, weight 2: broadcasts like the reference's Tensor arithmetic
    assert m.avg.shape == (2,)
    m.reset()
    assert float(m.sum) == 0.0 and m.count == 0


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.arange(6, dtype=torch.float32).reshape(2, 3))


class _Inferencer:
    def __init__(self, net):
        self.net = net
        self.modes = []

    def __call__(self, dataset):
        self.modes.append(self.net.training)
        return [dict(ap=a) for a in dataset]


class _Evaluator:
    metrics = ("AP", "AR")

    def __init__(self, seq):
        self.seq = list(seq)

    def __call__(self, result):
        return {"AP": self.seq.pop(0), "AR": 0.5}


def test_eval_callback_last_best_interval_and_summary(tmp_path):
    net = _Net().train()
    inf = _Inferencer(net)
    ev = _Evaluator([0.30, 0.20, 0.45])
    cb = EvalCallback(inf, ev, dataset=[1, 2], interval=2, max_epoch=5, save_best=True, save_last=True,
                      best_ckpt_path=str(tmp_path / "best.ckpt"), last_ckpt_path=str(tmp_path / "last.ckpt"), summary_dir=str(tmp_path))
    outs = []
    with cb:
        for epoch in range(1, 6):
            for step in range(3):
                cb.on_train_step_end(torch.tensor(1.0 / (epoch * (step + 1))))
            with torch.no_grad():
                net.w.add_(1.0)
            outs.append(cb.on_train_epoch_end(epoch, net, lr=1e-3 / epoch))
    # evaluation at epochs 2, 4 (interval) and 5 (max_epoch), with the net in eval mode, restored to train mode afterwards
    assert [bool(o) for o in outs] == [False, True, False, True, True]
    assert inf.modes == [False, False, False] and net.training
    # best = 0.45 at epoch 5; the epoch-4 result (0.20) did not overwrite the epoch-2 best
    assert cb.best_result == 0.45 and cb.best_epoch == 5
    best, last = load_checkpoint(str(tmp_path / "best.ckpt")), load_checkpoint(str(tmp_path / "last.ckpt"))
    assert np.array_equal(best["w"], net.w.detach().numpy()) and np.array_equal(last["w"], net.w.detach().numpy())
    recs = [json.loads(ln) for ln in open(tmp_path / "summary.jsonl")]
    assert [r["epoch"] for r in recs] == [1, 2, 3, 4, 5] and recs[-1]["step"] == 15
    assert abs(recs[0]["train/loss"] - np.mean([1.0, 0.5, 1 / 3])) < 1e-6          # the meter is reset every epoch
    assert "val/AP" in recs[1] and "val/AP" not in recs[0] and recs[4]["val/AP"] == 0.45


def test_eval_callback_without_evaluation_and_bad_metric(tmp_path):
    net = _Net()
    cb = EvalCallback(save_last=True, save_best=True, last_ckpt_path=str(tmp_path / "l.ckpt"), summary_dir=str(tmp_path))
    cb.on_train_step_end(0.5)
    assert cb.on_train_epoch_end(1, net, 1e-3) == {} and os.path.exists(tmp_path / "l.ckpt") and not os.path.exists("best.ckpt")
    with pytest.raises(ValueError, match="target metric"):
        EvalCallback(_Inferencer(net), _Evaluator([]), dataset=[1], target_metric_name="PCK")

    class _Boom(_Evaluator):
        def __call__(self, result):
            raise RuntimeError("no annotations")

    cb = EvalCallback(_Inferencer(net), _Boom([]), dataset=[1], save_best=True, best_ckpt_path=str(tmp_path / "b.ckpt"), summary_dir=str(tmp_path))
    cb.on_train_step_end(0.5)
    assert cb.on_train_epoch_end(1, net, 1e-3) == {} and not os.path.exists(tmp_path / "b.ckpt")  # training goes on, nothing saved


def _rank_main(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert float(Allreduce()(torch.tensor(float(rank + 1)))) == 3.0
        cb = EvalCallback(save_last=True, last_ckpt_path=os.path.join(out_dir, "last.ckpt"), summary_dir=out_dir, rank_id=rank, device_num=world)
        with cb:
            for v in (1.0, 3.0) if rank == 0 else (5.0, 7.0):
                cb.on_train_step_end(v)
            cb.on_train_epoch_end(1, _Net(), 1e-3)
        assert abs(float(cb.last_epoch_loss) - 4.0) < 1e-6  # mean over ranks of the per-rank epoch means (2 and 6)
    finally:
        dist.destroy_process_group()


def test_epoch_loss_is_averaged_over_ranks_world_size_2_gloo(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    tmp.spawn(_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    recs = [json.loads(ln) for ln in open(tmp_path / "summary.jsonl")]
    assert len(recs) == 1 and abs(recs[0]["train/loss"] - 4.0) < 1e-6  # only rank 0 writes the summary and the checkpoint
    assert os.path.exists(tmp_path / "last.ckpt")
