"""The conv tuner never communicates: a plan that only rank 0 builds - EvalCallback's rank-0 evaluation between
training epochs (callbacks/eval_callback.py, reference eval_callback.py:139-156) - must not strand or cross-match a collective.
World-size-2 gloo ranks: rank 0 alone walks the tuner for new and known keys while rank 1 is already inside the next gradient
all-reduce; the explicit share point (`share_tuner_choices` / `tune_on_rank0_first`) is the only collective."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp_

from mindpose_amd.models import layers


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.pop("MINDPOSE_TUNE_CACHE", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    layers._TUNE_CACHE.clear()

    def never(_v):  # the tuner's launch callback: small layers are not timed, so this is never called on CPU
        raise AssertionError("no trial launch expected")

    # 1. a "training epoch": both ranks use the same keys, rank 0 has tuned choices for them
    keys = [("conv", i, 64, 64, 3) for i in range(6)]
    if rank == 0:
        for i, k in enumerate(keys):
            layers._TUNE_CACHE[repr(k)] = 20 + i
    built = layers.tune_on_rank0_first(lambda: [layers._autotune(k, 1 << 30, 4, never) for k in keys])  # hits on rank 0 ...
    assert built == [20 + i for i in range(6)], built  # ... and, after the ONE broadcast, on rank 1
    # 2. rank 0 alone evaluates: new keys (small layers: heuristic, no timing) and known ones; rank 1 is already in the next
    #    step's all-reduce.  A collective inside the tuner would pair with that all-reduce: wrong sum or a hang.
    if rank == 0:
        for i in range(40):
            assert layers._autotune(("eval-only", i), 1 << 10, 4, never) == -1
            assert layers._autotune(keys[i % 6], 1 << 30, 4, never) == 20 + i % 6
    g = torch.full((8,), float(rank + 1))
    dist.all_reduce(g)
    assert torch.equal(g, torch.full((8,), 3.0)), g
    # 3. rank 0's rank-local choices stay rank-local until the next explicit share
    assert (repr(("eval-only", 0)) in layers._TUNE_CACHE) == (rank == 0)
    adopted = layers.share_tuner_choices()
    assert repr(("eval-only", 39)) in layers._TUNE_CACHE
    assert adopted == (0 if rank == 0 else 46)
    # 4. a failing build on rank 0 still reaches the broadcast: nobody hangs, rank 0 sees its own exception
    def build():
        if rank == 0:
            raise RuntimeError("out of memory (simulated)")
        return "built"
    try:
        res = layers.tune_on_rank0_first(build)
    except RuntimeError as exc:
        res = f"raised: {exc}"
    assert res == ("raised: out of memory (simulated)" if rank == 0 else "built")
    dist.barrier()
    with open(os.path.join(out_dir, f"ok_{rank}"), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


def test_rank0_only_plan_build_between_collectives_world_size_2_gloo(tmp_path):
    mp_.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(os.path.join(tmp_path, "ok_0")) and os.path.exists(os.path.join(tmp_path, "ok_1"))


def test_one_rank_is_a_plain_call():
    assert layers.tune_on_rank0_first(lambda: 7) == 7 and layers.share_tuner_choices() == 0
