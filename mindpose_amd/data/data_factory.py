"""``create_dataset`` / ``create_pipeline`` of the top-down path (reference: mindpose/data/data_factory.py:16-171), MI355X-first.

The reference wraps the record loader into ``mindspore.dataset.GeneratorDataset`` (shuffle when training, ``num_shards`` /
``shard_id`` sharding) and maps, per SAMPLE in CPU worker processes: Decode -> the transform list -> Normalize -> HWC2CHW ->
project -> batch.  Here the same steps run per BATCH with the pixel work on the GPU:

    decode (host, PIL)  ->  the transform list, sample by sample, exactly the reference's host geometry and random draws -
    but ``topdown_affine`` only records its matrix and ``topdown_generate_target`` only its key points  ->  ONE
    ``mp_warp_affine`` launch per batch (flip + warpAffine + Normalize + HWC2CHW fused, writes NCHW fp32) and ONE
    ``mp_gaussian_target`` launch  ->  project to the final columns.

Semantics kept: transform order and names (``register("transform")``), column names (``column_names.py``), normalisation
constants, ``drop_remainder=is_train``, sharding by ``device_num`` / ``rank_id``.
"""
import io
import logging
from typing import Any, Dict, Iterator, List, Optional, Sequence, Union

import numpy as np
import torch

from ..register import entrypoint
from .column_names import COLUMN_MAP, FINAL_COLUMN_MAP
from .transform.topdown_transform import TopDownAffine, TopDownGenerateTarget

__all__ = ["create_dataset", "create_pipeline", "ShardedDataset", "TopDownPipeline"]


class ShardedDataset:
    """The slice of ``GeneratorDataset(source, column_names, shuffle, num_shards, shard_id)`` the pipeline needs: an epoch's
    index order (a fresh permutation per epoch when ``shuffle``), dealt round-robin to the shards - index k of shard s is entry
    ``s + k * num_shards`` of the order, wrapped so that every shard has ceil(len / num_shards) samples [MS-knowledge:
    DistributedSampler pads with the head of the order]."""

    def __init__(self, source, column_names: Sequence[str], shuffle: bool, num_shards: Optional[int] = None,
                 shard_id: Optional[int] = None, num_parallel_workers: int = 1, seed: int = 0) -> None:
        if (num_shards is None) != (shard_id is None):
            raise ValueError("num_shards and shard_id must be given together")
        if num_shards is not None and not 0 <= shard_id < num_shards:
            raise ValueError(f"shard_id {shard_id} outside [0, {num_shards})")
        self.source, self.column_names, self.shuffle = source, list(column_names), shuffle
        self.num_shards, self.shard_id = num_shards or 1, shard_id or 0
        self.num_parallel_workers = num_parallel_workers
        self.seed, self.epoch = seed, 0

    def __len__(self) -> int:
        return (len(self.source) + self.num_shards - 1) // self.num_shards

    def get_dataset_size(self) -> int:
        return len(self)

    def indices(self) -> np.ndarray:
        n = len(self.source)
        order = np.random.RandomState(self.seed + self.epoch).permutation(n) if self.shuffle else np.arange(n)
        picks = (self.shard_id + np.arange(len(self)) * self.num_shards) % max(n, 1)
        return order[picks] if n else order

    def __iter__(self) -> Iterator[Dict[str, Any]]:
        for i in self.indices():
            yield dict(zip(self.column_names, self.source[int(i)]))
        self.epoch += 1


def create_dataset(image_root: str, annotation_file: Optional[str] = None, dataset_format: str = "coco_topdown",
                   is_train: bool = True, device_num: Optional[int] = None, rank_id: Optional[int] = None, num_workers: int = 1,
                   config: Optional[Dict[str, Any]] = None, **kwargs: Any) -> ShardedDataset:
    """Signature and behaviour of data_factory.py:16-68: ``None`` keyword arguments are dropped, the dataset class is looked up in
    the registry, training data is shuffled, every rank reads shard ``rank_id`` of ``device_num``."""
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    dataset = entrypoint("dataset", dataset_format)(image_root, annotation_file, is_train=is_train, config=config, **kwargs)
    column_names = COLUMN_MAP[dataset_format]["train" if is_train else "val"]
    return ShardedDataset(dataset, column_names=column_names, shuffle=is_train, num_shards=device_num, shard_id=rank_id,
                          num_parallel_workers=num_workers)


def _decode(data) -> np.ndarray:
    """``vision.Decode()``: encoded bytes -> RGB uint8 [H, W, 3].  ``.npy`` payloads (synthetic fixtures) load as they are."""
    buf = data.tobytes() if isinstance(data, np.ndarray) else bytes(data)
    if buf[:6] == b"\x93NUMPY":
        arr = np.load(io.BytesIO(buf), allow_pickle=False)
    else:
        from PIL import Image  # the image codec is host plumbing
        with Image.open(io.BytesIO(buf)) as im:
            arr = np.asarray(im.convert("RGB"))
    if arr.ndim != 3 or arr.shape[2] != 3 or arr.dtype != np.uint8:
        raise ValueError(f"decoded image must be uint8 [H, W, 3], got {arr.dtype} {arr.shape}")
    return arr


class TopDownPipeline:
    """What ``create_pipeline`` returns: iterating yields batches - dicts with the final columns of ``column_names.py``;
    ``image`` ([B, 3, H, W] fp32 normalised), ``target`` / ``target_weight``, ``center`` / ``scale`` / ``bbox_scores`` are CUDA
    tensors, ``boxes`` / ``bbox_ids`` numpy, ``image_file`` a list (what ``TopDownHeatMapInferencer.infer`` and the training
    step consume)."""

    def __init__(self, dataset: ShardedDataset, transforms: list, column_names: List[str], final_column_names: List[str],
                 batch_size: int, is_train: bool, normalize: bool, normalize_mean, normalize_std, hwc_to_chw: bool, num_workers: int,
                 device: Optional[torch.device] = None) -> None:
        self.dataset, self.transforms = dataset, transforms
        self.column_names, self.final_column_names = column_names, final_column_names
        self.batch_size, self.is_train = int(batch_size), is_train
        self.normalize, self.mean, self.std, self.hwc_to_chw = normalize, tuple(normalize_mean), tuple(normalize_std), hwc_to_chw
        self.num_workers = num_workers
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.batch_size < 1:
            raise ValueError("batch_size must be >= 1")

    def __len__(self) -> int:
        n = len(self.dataset)
        return n // self.batch_size if self.is_train else (n + self.batch_size - 1) // self.batch_size

    def get_dataset_size(self) -> int:
        return len(self)

    def create_dict_iterator(self, num_epochs: int = 1):
        for _ in range(num_epochs):
            yield from iter(self)

    # -- one sample through the transform list (host part) -----------------------------------------------------------------------
    def _run_sample(self, state: Dict[str, Any]) -> Dict[str, Any]:
        if not (isinstance(state["image"], np.ndarray) and state["image"].ndim == 3):  # not decoded ahead by the thread pool
            state["image"] = _decode(state["image"])
        for t in self.transforms:
            if isinstance(t, TopDownAffine):  # matrix + key points now, pixels with the batch
                trans = t.get_matrix(state["center"], state["scale"], state["rotation"])
                state["_affine"], state["_trans"] = t, trans
                if "keypoints" in state:
                    state["keypoints"] = t.transform_keypoints(state["keypoints"], trans)
            elif isinstance(t, TopDownGenerateTarget):  # key points now, heat maps with the batch
                state["_target"] = t
            else:
                state.update(t.transform(state))
        return state

    def _finish(self, states: List[Dict[str, Any]]) -> Dict[str, Any]:
        dev = self.device
        out: Dict[str, Any] = {}
        aff = states[0].get("_affine")
        if aff is not None:
            images, flips = [], []
            for s in states:
                im = s["image"]
                flipped = im.strides[1] < 0  # topdown_horizontal_random_flip hands back a mirrored VIEW: the kernel mirrors while sampling
                images.append(torch.from_numpy(np.ascontiguousarray(im[:, ::-1] if flipped else im)).to(dev, non_blocking=True))
                flips.append(flipped)
            mats = np.stack([s["_trans"] for s in states])
            fused = self.normalize and self.hwc_to_chw
            crops = aff._launch(images, list(range(len(states))), mats, fused, None, self.mean, self.std, flips if any(flips) else None)
            if not fused:  # rare combinations: the warped uint8 image, then the requested steps as tensor ops
                crops = crops.float()
                if self.normalize:
                    crops = (crops - torch.tensor(self.mean, device=dev) * 255.0) / (torch.tensor(self.std, device=dev) * 255.0)
                if self.hwc_to_chw:
                    crops = crops.permute(0, 3, 1, 2).contiguous()
            out["image"] = crops
        else:
            raise ValueError("the top-down pipeline needs `topdown_affine` in its transform list (fixed-size network input)")
        tgt = states[0].get("_target")
        if tgt is not None:
            kp = torch.from_numpy(np.stack([np.asarray(s["keypoints"], dtype=np.float32) for s in states])).to(dev)
            out["target"], out["target_weight"] = tgt.generate(kp)
        for name in self.final_column_names:
            if name in out:
                continue
            vals = [s[name] for s in states]
            if name in ("center", "scale", "bbox_scores"):
                out[name] = torch.from_numpy(np.stack([np.asarray(v, dtype=np.float32) for v in vals])).to(dev)
            elif name == "image_file":
                out[name] = [str(v) for v in vals]
            else:
                out[name] = np.stack([np.asarray(v) for v in vals])
        return {k: out[k] for k in self.final_column_names}

    def __iter__(self) -> Iterator[Dict[str, Any]]:
        pool = None
        if self.num_workers > 1:
            from concurrent.futures import ThreadPoolExecutor  # decoding releases the GIL; the random draws stay on this thread
            pool = ThreadPoolExecutor(self.num_workers)
        try:
            pending: List[Dict[str, Any]] = []
            for state in self.dataset:
                pending.append(state)
                if len(pending) == self.batch_size:
                    yield self._batch(pending, pool)
                    pending = []
            if pending and not self.is_train:  # drop_remainder = is_train (data_factory.py:148-150)
                yield self._batch(pending, pool)
        finally:
            if pool is not None:
                pool.shutdown()

    def _batch(self, pending: List[Dict[str, Any]], pool) -> Dict[str, Any]:
        if pool is not None:  # decode in parallel; the transform list (with its np.random draws) still runs in sample order
            for s, im in zip(pending, pool.map(lambda st: _decode(st["image"]), pending)):
                s["image"] = im
        return self._finish([self._run_sample(dict(s)) for s in pending])


def create_pipeline(dataset: ShardedDataset, transforms: List[Union[str, Dict[str, Any]]], method: str = "topdown", batch_size: int = 1,
                    is_train: bool = True, normalize: bool = True, normalize_mean: List[float] = [0.485, 0.456, 0.406],
                    normalize_std: List[float] = [0.229, 0.224, 0.255], hwc_to_chw: bool = True, num_workers: int = 1,
                    config: Optional[Dict[str, Any]] = None) -> TopDownPipeline:
    """Signature of data_factory.py:71-151 (the reference's ``normalize_std`` really ends in 0.255: checkpoints were trained with it)."""
    if method not in FINAL_COLUMN_MAP:
        raise ValueError(f"method `{method}` is outside the top-down hot path (supported: {sorted(FINAL_COLUMN_MAP)})")
    key = "train" if is_train else "val"
    column_names, final_column_names = COLUMN_MAP[method][key], FINAL_COLUMN_MAP[method][key]
    transform_funcs = _convert_names_to_transform(transforms, is_train=is_train, config=config)
    logging.info(f"pipeline: {[type(t).__name__ for t in transform_funcs]}, batch {batch_size}")
    return TopDownPipeline(dataset, transform_funcs, column_names, final_column_names, batch_size, is_train, normalize, normalize_mean,
                           normalize_std, hwc_to_chw, num_workers)


def _convert_names_to_transform(names_with_args: List[Union[str, Dict[str, Any]]], is_train: bool = True,
                                config: Optional[Dict[str, Any]] = None) -> list:
    """A list of names, or one-entry dicts ``{name: kwargs}``, to transform objects (data_factory.py:154-171)."""
    transforms = list()
    for name_with_arg in names_with_args:
        if isinstance(name_with_arg, str):
            name, kwargs = name_with_arg, dict()
        else:
            name = list(name_with_arg.keys())[0]
            kwargs = list(name_with_arg.values())[0] or dict()
        transforms.append(entrypoint("transform", name)(is_train=is_train, config=config, **kwargs))
    return transforms
