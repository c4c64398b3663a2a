"""Evaluation / checkpoint callback of the training loop, mirroring mindpose/callbacks/eval_callback.py:16-202: per-step loss
meter, per-epoch log line, mean loss over the ranks, ``last.ckpt`` every epoch, evaluation every ``interval`` epochs (and at
``max_epoch``) on rank 0, ``best.ckpt`` when the target metric improves, scalar records for plotting.

There is no ``mindspore.Model`` driving it here: the training loop calls ``on_train_step_end(loss)`` and
``on_train_epoch_end(cur_epoch, network, lr)`` itself (``tools/train_smoke.py`` style loops).  The scalar records the reference
writes through ``SummaryRecord`` go to ``<summary_dir>/summary.jsonl`` (one JSON object per epoch: step, epoch, train/loss,
train/lr, val/<metric>...)."""
import json
import logging
import os
from typing import Any, Dict, Optional

import torch

from ..utils.ckpt import save_checkpoint
from ..utils.misc import Allreduce, AverageMeter


class EvalCallback:
    def __init__(self, inferencer=None, evaluator=None, dataset=None, interval: int = 1, max_epoch: int = 1, save_best: bool = False,
                 save_last: bool = False, best_ckpt_path: str = "./best.ckpt", last_ckpt_path: str = "./last.ckpt",
                 target_metric_name: str = "AP", summary_dir: str = ".", rank_id: Optional[int] = None,
                 device_num: Optional[int] = None) -> None:
        self.inferencer, self.evaluator, self.dataset = inferencer, evaluator, dataset
        self.interval, self.max_epoch = interval, max_epoch
        self.save_best, self.save_last = save_best, save_last
        self.best_ckpt_path, self.last_ckpt_path = os.path.abspath(best_ckpt_path), os.path.abspath(last_ckpt_path)
        self.target_metric_name = target_metric_name
        self.summary_dir = summary_dir
        self.rank_id = rank_id if rank_id is not None else 0
        self.device_num = device_num if device_num is not None else 1
        self._eval_during_train = not (inferencer is None or evaluator is None or dataset is None)
        if not self._eval_during_train:
            logging.info("Evaluation during training is disabled.")
            if save_best:
                logging.warning("Best model cannot be saved since `val_while_train` is disabled.")
        elif target_metric_name not in evaluator.metrics:
            raise ValueError(f"target metric `{target_metric_name}` is not listed in evaluator metrics `{evaluator.metrics}`")
        self.all_reduce = Allreduce() if self.device_num > 1 else None
        self.best_result, self.best_epoch = 0.0, 0
        self.loss_meter = AverageMeter()
        self.cur_step = 0
        self._summary = None

    # -- summary file (rank 0) ----------------------------------------------------------------------------------------------
    def __enter__(self):
        if self.rank_id == 0:
            os.makedirs(self.summary_dir, exist_ok=True)
            self._summary = open(os.path.join(self.summary_dir, "summary.jsonl"), "a")
        return self

    def __exit__(self, *err):
        if self._summary is not None:
            self._summary.close()
            self._summary = None

    # -- hooks --------------------------------------------------------------------------------------------------------------
    def on_train_step_end(self, loss) -> None:
        self.cur_step += 1
        self.loss_meter.update(loss)

    def on_train_epoch_end(self, cur_epoch: int, network, lr: float) -> Dict[str, Any]:
        """Returns the evaluation output of this epoch ({} when none ran)."""
        avg = self.loss_meter.avg
        logging.info(f"[rank = {self.rank_id}] epoch = {cur_epoch}, lr = {float(lr):.3e}, loss = {float(avg.sum()):.6f}")
        loss_avg = avg
        if self.all_reduce is not None:  # mean over the ranks of the per-rank epoch mean
            dev = next(network.parameters()).device if hasattr(network, "parameters") else torch.device("cpu")
            loss_avg = (self.all_reduce(avg.to(dev, torch.float32)) / self.device_num).to("cpu", torch.float64)
        self.loss_meter.reset()
        if self.rank_id == 0 and self.save_last:
            self._save(network, self.last_ckpt_path)
            logging.info(f"Last checkpoint is saved at `{self.last_ckpt_path}`.")
        output: Dict[str, Any] = {}
        if (cur_epoch % self.interval == 0 or cur_epoch == self.max_epoch) and self.rank_id == 0 and self._eval_during_train:
            was_training = getattr(self.inferencer.net, "training", False)
            self.inferencer.net.eval()  # the inferencer must not run in training mode
            try:
                result = self.inferencer(self.dataset)
                output = self.evaluator(result)
                logging.info(output)
                if self.save_best:
                    self._save_best_model(network, float(output[self.target_metric_name]), cur_epoch)
            except Exception as e:  # the reference keeps training when an evaluation fails
                logging.warning(f"Error occured at evaluation. {e}")
                output = {}
            finally:
                if was_training:
                    self.inferencer.net.train()
        if self.rank_id == 0 and self._summary is not None:
            rec = {"step": self.cur_step, "epoch": cur_epoch, "train/lr": float(lr), "train/loss": float(loss_avg.sum())}
            if loss_avg.numel() > 1:
                rec.update({f"train/loss_{i}": float(v) for i, v in enumerate(loss_avg.flatten())})
            rec.update({"val/" + k: float(v) for k, v in output.items()})
            self._summary.write(json.dumps(rec) + "\n")
            self._summary.flush()
        self.last_epoch_loss = loss_avg
        return output

    # -- checkpoints ----------------------------------------------------------------------------------------------------------
    @staticmethod
    def _save(network, path: str) -> None:
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        save_checkpoint({k: v.detach().cpu().numpy() for k, v in network.state_dict().items()}, path)

    def _save_best_model(self, network, result: float, cur_epoch: int) -> None:
        logging.info(f"epoch: {cur_epoch}, current result: {result:.3f}, previous_best_result: {self.best_result:.3f}.")
        if result > self.best_result:
            self.best_result, self.best_epoch = result, cur_epoch
            self._save(network, self.best_ckpt_path)
            logging.info(f"Best result is {self.best_result:.3f} at {self.best_epoch} epoch. "
                         f"Best checkpoint is saved at `{self.best_ckpt_path}`.")
        else:
            logging.info(f"Best result is {self.best_result:.3f} at {self.best_epoch} epoch. Best checkpoint is unchanged.")
