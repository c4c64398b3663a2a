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
import atexit
import io
import logging
import os
import queue
import struct
import subprocess
import sys
import threading
import weakref
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
    if isinstance(data, str):  # dataset.topdown.ImagePath
        data = np.fromfile(str(data), dtype=np.uint8)
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


class _DecodeProcesses:
    """``workers`` codec processes (data/decode_worker.py: numpy + PIL only, started as scripts - no fork of this GPU-initialised
    process, no GPU in the children) decoding into one shared-memory block of ``regions x batch`` slots.  `submit` hands a batch's
    encoded payloads to the workers round-robin from a feeder thread and returns at once; `collect` waits for the replies and
    returns the decoded images as VIEWS of the block (None where an image did not fit its slot: the caller decodes that one
    itself).  A region is reused ``regions`` submits later: the caller copies a batch out before then (`_finish` packs it into its
    pinned upload buffer)."""

    def __init__(self, workers: int, batch: int, slot_bytes: int, regions: int = 3) -> None:
        from multiprocessing import shared_memory
        self.batch, self.slot_bytes, self.regions = batch, slot_bytes, regions
        self.shm = shared_memory.SharedMemory(create=True, size=regions * batch * slot_bytes)
        script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "decode_worker.py")
        self.procs = [subprocess.Popen([sys.executable, script, self.shm.name, str(slot_bytes)], stdin=subprocess.PIPE, stdout=subprocess.PIPE)
                      for _ in range(max(1, workers))]
        self._region = 0
        self._closed = False
        # The block is page-locked for the GPU (hipHostRegister) when there is one: a decoded image then goes from its slot to the
        # device by ONE asynchronous DMA - no second host copy into a staging tensor (118 MB per batch of 128 VGA images).  A region
        # is handed to the workers again only after the uploads that read it have completed (`consumed`).
        self.pinned = False
        self._consumed: Dict[int, Any] = {}
        if torch.cuda.is_available():
            import ctypes
            self._addr = ctypes.addressof(ctypes.c_char.from_buffer(self.shm.buf))
            try:
                self.pinned = int(torch.cuda.cudart().cudaHostRegister(self._addr, self.shm.size, 0)) == 0
            except Exception:  # noqa: BLE001 - not fatal: the upload goes through a pinned staging tensor instead
                self.pinned = False
        # ONE feeder thread writes the requests of every submit, in submit order: two submits in flight (the look-ahead of the
        # pipeline) must not interleave their frames on a worker's pipe
        self._feed_q: "queue.Queue" = queue.Queue()
        self._feeder = threading.Thread(target=self._feed_loop, name="mindpose-loader-feed", daemon=True)
        self._feeder.start()
        self._finalizer = weakref.finalize(self, _DecodeProcesses._shutdown, self.procs, self.shm)
        atexit.register(self._finalizer)

    def consumed(self, region: int, event) -> None:
        """``event``: recorded behind the last device copy that reads ``region``."""
        self._consumed[region] = event

    def submit(self, payloads: Sequence[Any]):
        region, self._region = self._region, (self._region + 1) % self.regions
        ev = self._consumed.pop(region, None)
        if ev is not None:
            ev.synchronize()  # (two submits ago: long done in practice)
        n, nw = len(payloads), len(self.procs)
        if n > self.batch:
            raise ValueError("more payloads than slots in a region")

        done = threading.Event()
        self._feed_q.put((region, list(payloads), done))
        return region, n, done

    def _feed_loop(self) -> None:
        nw = len(self.procs)
        while True:
            job = self._feed_q.get()
            if job is None:
                return
            region, payloads, done = job
            try:
                for i, data in enumerate(payloads):
                    pipe = self.procs[i % nw].stdin
                    if isinstance(data, str):  # dataset.topdown.ImagePath: the worker reads the file
                        buf = str(data).encode("utf-8")
                        pipe.write(struct.pack("<qq", region * self.batch + i, -len(buf)))
                    else:
                        buf = memoryview(np.ascontiguousarray(data)).cast("B") if isinstance(data, np.ndarray) else bytes(data)
                        pipe.write(struct.pack("<qq", region * self.batch + i, len(buf)))
                    pipe.write(buf)
                    pipe.flush()
            except (OSError, ValueError):  # a worker went away: `collect` reports it (short read on its stdout)
                pass
            finally:
                done.set()

    def collect(self, ticket) -> List[Optional[np.ndarray]]:
        region, n, fed = ticket
        nw = len(self.procs)
        images: List[Optional[np.ndarray]] = [None] * n
        for w, proc in enumerate(self.procs):
            for i in range(w, n, nw):  # a worker answers its requests in order
                reply = proc.stdout.read(24)
                if len(reply) < 24:
                    raise RuntimeError(f"image-codec worker {w} ended (exit code {proc.poll()})")
                slot, h, wd = struct.unpack("<qqq", reply)
                if slot != region * self.batch + i:
                    raise RuntimeError("image-codec worker answered out of order")
                if h == -2:
                    raise ValueError(f"sample {i} of the batch could not be decoded to uint8 [H, W, 3]")
                if h >= 0:
                    images[i] = np.ndarray((h, wd, 3), np.uint8, buffer=self.shm.buf, offset=slot * self.slot_bytes)
        fed.wait()
        self.last_region = region
        return images

    @staticmethod
    def _shutdown(procs, shm) -> None:
        for p in procs:
            try:
                p.stdin.write(struct.pack("<qq", 0, 0))
                p.stdin.flush()
                p.stdin.close()
            except (OSError, ValueError):
                pass
        for p in procs:
            try:
                p.wait(timeout=2)
            except subprocess.TimeoutExpired:
                p.kill()
            try:
                p.stdout.close()
            except (OSError, ValueError):
                pass
        try:
            shm.close()
            shm.unlink()
        except (OSError, FileNotFoundError):
            pass

    def close(self) -> None:
        if not self._closed:
            self._closed = True
            if self.pinned:
                try:
                    torch.cuda.synchronize()
                    torch.cuda.cudart().cudaHostUnregister(self._addr)
                except Exception:  # noqa: BLE001
                    pass
            self._feed_q.put(None)
            self._feeder.join(timeout=2)
            atexit.unregister(self._finalizer)
            self._finalizer()


class TopDownPipeline:
    """What ``create_pipeline`` returns: iterating yields batches - dicts with the final columns of ``column_names.py``;
    ``image`` ([B, 3, H, W] fp32 normalised), ``target`` / ``target_weight``, ``center`` / ``scale`` / ``bbox_scores`` are CUDA
    tensors, ``boxes`` / ``bbox_ids`` numpy, ``image_file`` a list (what ``TopDownHeatMapInferencer.infer`` and the training
    step consume)."""

    def __init__(self, dataset: ShardedDataset, transforms: list, column_names: List[str], final_column_names: List[str],
                 batch_size: int, is_train: bool, normalize: bool, normalize_mean, normalize_std, hwc_to_chw: bool, num_workers: int,
                 device: Optional[torch.device] = None, prefetch: Optional[int] = None) -> None:
        """``prefetch``: batches prepared AHEAD of the consumer by a background thread on a side stream (decode in the thread pool,
        one pinned-memory upload, crop + target launches) while the training step of the previous batch runs; 0 = the synchronous
        form.  None = ``MINDPOSE_LOADER_PREFETCH`` (default 2).  The samples, their order and the order of the random draws are
        those of the synchronous form: the batches are bit-equal (tests/test_gpu_loader.py)."""
        self.dataset, self.transforms = dataset, transforms
        self.prefetch = int(os.environ.get("MINDPOSE_LOADER_PREFETCH", "2")) if prefetch is None else int(prefetch)
        # num_workers > 1: the image codec runs in that many worker PROCESSES (`_DecodeProcesses`; MINDPOSE_LOADER_DECODE=thread keeps
        # round 4's thread pool - PIL holds the interpreter lock for most of a decode, so threads stop scaling at ~2x one core)
        self.decode_mode = os.environ.get("MINDPOSE_LOADER_DECODE", "process")
        self.slot_bytes = int(float(os.environ.get("MINDPOSE_LOADER_SLOT_MB", "2")) * (1 << 20))  # 2 MB: 832 x 832 x 3 (COCO: <= 640 x 640)
        self._codec: Optional[_DecodeProcesses] = None
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

    def _finish(self, states: List[Dict[str, Any]], pool=None, codec=None, region=None) -> Dict[str, Any]:
        dev = self.device
        out: Dict[str, Any] = {}
        aff = states[0].get("_affine")
        if aff is not None:
            # ONE upload per batch: the decoded images packed into a pinned staging tensor (torch's caching host allocator hands the
            # block out again only after this stream has passed the copy), one asynchronous copy, the kernel's sources = views of it.
            # (Per-image pageable copies were 128 synchronous transfers of ~0.9 MB per batch.)
            srcs, flips, sizes = [], [], []
            for s in states:
                im = s["image"]
                flipped = im.strides[1] < 0  # topdown_horizontal_random_flip hands back a mirrored VIEW: the kernel mirrors while sampling
                srcs.append(im[:, ::-1] if flipped else im)
                flips.append(flipped)
                sizes.append((im.size + 255) & ~255)  # 256-byte aligned slots
            offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).tolist()
            if codec is not None:
                # the decoded images sit in page-locked shared memory (`_DecodeProcesses`): one asynchronous DMA per image, slot -> device
                packed = torch.empty(sum(sizes), dtype=torch.uint8, device=dev)
                for i, im in enumerate(srcs):
                    flat = torch.from_numpy(im.reshape(-1)) if im.flags["C_CONTIGUOUS"] else torch.from_numpy(np.ascontiguousarray(im).reshape(-1))
                    packed[offsets[i]:offsets[i] + im.size].copy_(flat, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                codec.consumed(region, ev)
            else:
                stage = torch.empty(sum(sizes), dtype=torch.uint8, pin_memory=True)
                host = stage.numpy()

                def pack(i):
                    np.copyto(host[offsets[i]:offsets[i] + srcs[i].size].reshape(srcs[i].shape), srcs[i])
                if pool is not None:
                    list(pool.map(pack, range(len(srcs))))  # ~118 MB per batch of 128 VGA images: the copies run on the pool's threads
                else:
                    for i in range(len(srcs)):
                        pack(i)
                packed = stage.to(dev, non_blocking=True)
            images, off = [], 0
            for im, size in zip(srcs, sizes):
                images.append(packed[off:off + im.size].view(im.shape))
                off += size
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
        if self.prefetch > 0 and self.device.type == "cuda":
            yield from self._iter_prefetched()
        else:
            yield from self._iter_batches()

    def _iter_prefetched(self) -> Iterator[Dict[str, Any]]:
        """The batches of `_iter_batches`, prepared up to ``prefetch`` ahead by ONE background thread (so the samples and the global
        ``np.random`` draws keep their order) on a side stream; the consumer's stream waits for the batch's event and takes over
        its tensors (``record_stream``: they were allocated on the side stream's pool)."""
        dev = self.device
        side = torch.cuda.Stream(device=dev)
        q: "queue.Queue" = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()

        def produce():
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(side):
                    for batch in self._iter_batches():
                        ev = torch.cuda.Event()
                        ev.record(side)
                        while not stop.is_set():
                            try:
                                q.put((batch, ev), timeout=0.1)
                                break
                            except queue.Full:
                                continue
                        if stop.is_set():
                            return
                q.put((None, None))
            except BaseException as exc:  # noqa: BLE001 - handed to the consumer, which re-raises it
                q.put((exc, None))

        worker = threading.Thread(target=produce, name="mindpose-loader-prefetch", daemon=True)
        worker.start()
        try:
            while True:
                batch, ev = q.get()
                if batch is None:
                    return
                if isinstance(batch, BaseException):
                    raise batch
                cur = torch.cuda.current_stream(dev)
                cur.wait_event(ev)
                for v in batch.values():
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(cur)
                yield batch
        finally:
            stop.set()
            while worker.is_alive():  # a producer blocked on a full queue sees the flag within its timeout
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                worker.join(timeout=0.05)

    def close(self) -> None:
        """Stop the codec processes and release their shared-memory block (also done when the pipeline is collected / at exit)."""
        if self._codec is not None:
            self._codec.close()
            self._codec = None

    def _iter_batches(self) -> Iterator[Dict[str, Any]]:
        pool = None
        if self.num_workers > 1:
            from concurrent.futures import ThreadPoolExecutor  # (thread mode: the codec; both modes: the copies into the upload buffer)
            pool = ThreadPoolExecutor(self.num_workers)
            if self.decode_mode == "process" and self._codec is None:
                self._codec = _DecodeProcesses(self.num_workers, self.batch_size, self.slot_bytes)
                if hasattr(self.dataset.source, "_column_sources"):  # TopDownDataset: hand out paths, the workers read the files
                    self.dataset.source.lazy_image = True
        codec = self._codec if (self.num_workers > 1 and self.decode_mode == "process") else None
        try:
            # the decodes of batch k + 1 are handed to the workers BEFORE batch k's transform list runs on this thread: the two stages
            # (image codec in the workers, per-sample geometry / draws here) overlap instead of alternating
            ahead = None
            for group in self._groups():
                if codec is not None:
                    ticket = codec.submit([st["image"] for st in group])
                elif pool is not None:
                    ticket = [pool.submit(_decode, st["image"]) for st in group]
                else:
                    ticket = None
                if ahead is not None:
                    yield self._batch(*ahead, pool, codec)
                ahead = (group, ticket)
            if ahead is not None:
                yield self._batch(*ahead, pool, codec)
        finally:
            if pool is not None:
                pool.shutdown()

    def _groups(self) -> Iterator[List[Dict[str, Any]]]:
        pending: List[Dict[str, Any]] = []
        for state in self.dataset:
            pending.append(state)
            if len(pending) == self.batch_size:
                yield pending
                pending = []
        if pending and not self.is_train:  # drop_remainder = is_train (data_factory.py:148-150)
            yield pending

    def _batch(self, pending: List[Dict[str, Any]], ticket, pool, codec) -> Dict[str, Any]:
        # decoded in parallel; the transform list (with its np.random draws) still runs in sample order on this thread
        region = None
        if codec is not None:
            for s, im in zip(pending, codec.collect(ticket)):
                s["image"] = im if im is not None else _decode(s["image"])  # (an image larger than a slot: decoded here)
            region = codec.last_region if codec.pinned else None
        elif ticket is not None:
            for s, f in zip(pending, ticket):
                s["image"] = f.result()
        return self._finish([self._run_sample(dict(s)) for s in pending], pool, codec if region is not None else None, region)


def create_pipeline(dataset: ShardedDataset, transforms: List[Union[str, Dict[str, Any]]], method: str = "topdown", batch_size: int = 1,
                    is_train: bool = True, normalize: bool = True, normalize_mean: List[float] = [0.485, 0.456, 0.406],
                    normalize_std: List[float] = [0.229, 0.224, 0.255], hwc_to_chw: bool = True, num_workers: int = 1,
                    config: Optional[Dict[str, Any]] = None, prefetch: Optional[int] = None) -> TopDownPipeline:
    """Signature of data_factory.py:71-151 (the reference's ``normalize_std`` really ends in 0.255: checkpoints were trained with it)
    plus ``prefetch`` (batches prepared ahead on a side stream, `TopDownPipeline`; MindSpore's dataset engine prefetches by itself)."""
    if method not in FINAL_COLUMN_MAP:
        raise ValueError(f"method `{method}` is outside the top-down hot path (supported: {sorted(FINAL_COLUMN_MAP)})")
    key = "train" if is_train else "val"
    column_names, final_column_names = COLUMN_MAP[method][key], FINAL_COLUMN_MAP[method][key]
    transform_funcs = _convert_names_to_transform(transforms, is_train=is_train, config=config)
    logging.info(f"pipeline: {[type(t).__name__ for t in transform_funcs]}, batch {batch_size}")
    return TopDownPipeline(dataset, transform_funcs, column_names, final_column_names, batch_size, is_train, normalize, normalize_mean,
                           normalize_std, hwc_to_chw, num_workers, prefetch=prefetch)


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
