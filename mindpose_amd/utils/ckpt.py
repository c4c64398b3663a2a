"""MindSpore ``.ckpt`` reader / writer and ``load_param_into_net`` without MindSpore.

Replaces ``mindspore.load_checkpoint`` + ``mindspore.load_param_into_net`` as the reference calls them
(tools/eval.py:64-68, tools/train.py:163-167, mindpose/models/backbones/utils.py:40-57), so that a released mindpose
checkpoint (e.g. ``hrnet_w32_256_192.ckpt``, configs/hrnet/README.md:17) loads straight into the networks of this
package: the parameter names are the reference's (``backbone.stage2.0.branches.0.0.bn1.moving_variance``, ...).

File format [MS-knowledge: mindspore/ccsrc/utils/checkpoint.proto, proto2; PARITY UNPINNED - no checkpoint file or
MindSpore install is available here, the wire format below is restated from that schema and pinned only by a
hand-assembled byte fixture in tests/test_ckpt_cpu.py]:

    message Checkpoint { repeated Value value = 1; }
    message Value      { required string tag = 1; required TensorProto tensor = 2; }
    message TensorProto{ repeated int64 dims = 1; required string tensor_type = 2; required bytes tensor_content = 3; }

A file is one or more serialised ``Checkpoint`` messages back to back (MindSpore flushes in slices), which protobuf
semantics merge into one; a tensor split over several ``Value`` entries with the same tag is concatenated.
"""
from typing import Dict, Iterable, List, Tuple

import numpy as np

_TYPES = {
    "Float32": np.float32, "Float16": np.float16, "Float64": np.float64, "Int8": np.int8, "Int16": np.int16,
    "Int32": np.int32, "Int64": np.int64, "UInt8": np.uint8, "UInt16": np.uint16, "UInt32": np.uint32,
    "UInt64": np.uint64, "Bool": np.bool_,
}
_NAMES = {np.dtype(v): k for k, v in _TYPES.items()}


class CheckpointFormatError(ValueError):
    pass


def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out = shift = 0
    while True:
        if pos >= len(buf):
            raise CheckpointFormatError("truncated varint")
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 70:
            raise CheckpointFormatError("varint too long")


def _fields(buf: bytes) -> Iterable[Tuple[int, int, object]]:
    """Yield (field number, wire type, value) of one message; value = int (varint) or memoryview (length-delimited)."""
    view = memoryview(buf)
    pos = 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            n, pos = _varint(buf, pos)
            if pos + n > len(buf):
                raise CheckpointFormatError("truncated length-delimited field")
            v = view[pos:pos + n]
            pos += n
        elif wt == 1:
            v, pos = view[pos:pos + 8], pos + 8
        elif wt == 5:
            v, pos = view[pos:pos + 4], pos + 4
        else:
            raise CheckpointFormatError(f"unsupported wire type {wt}")
        yield num, wt, v


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _tensor(buf: bytes):
    dims: List[int] = []
    ttype, content = None, b""
    for num, wt, v in _fields(buf):
        if num == 1 and wt == 0:
            dims.append(_signed64(v))
        elif num == 1 and wt == 2:  # packed encoding
            b, pos = bytes(v), 0
            while pos < len(b):
                d, pos = _varint(b, pos)
                dims.append(_signed64(d))
        elif num == 2 and wt == 2:
            ttype = bytes(v).decode("utf-8")
        elif num == 3 and wt == 2:
            content = bytes(v)
    if ttype is None:
        raise CheckpointFormatError("TensorProto without tensor_type")
    return dims, ttype, content


def load_checkpoint(ckpt_file_name: str, filter_prefix=None) -> Dict[str, np.ndarray]:
    """``mindspore.load_checkpoint(ckpt_file_name)`` -> ``{parameter name: numpy array}`` (insertion order = file order).

    ``filter_prefix``: a prefix or list of prefixes to drop (e.g. ``["moment1.", "moment2."]``), as in MindSpore."""
    with open(ckpt_file_name, "rb") as f:
        data = f.read()
    if not data:
        raise CheckpointFormatError(f"{ckpt_file_name} is empty")
    prefixes = [filter_prefix] if isinstance(filter_prefix, str) else list(filter_prefix or [])
    parts: Dict[str, list] = {}
    for num, wt, v in _fields(data):
        if num != 1 or wt != 2:
            continue  # unknown top-level fields are skipped, as protobuf does
        tag, tensor = None, None
        for n2, w2, v2 in _fields(bytes(v)):
            if n2 == 1 and w2 == 2:
                tag = bytes(v2).decode("utf-8")
            elif n2 == 2 and w2 == 2:
                tensor = _tensor(bytes(v2))
        if tag is None or tensor is None:
            raise CheckpointFormatError("checkpoint Value without tag or tensor")
        if any(tag.startswith(p) for p in prefixes):
            continue
        parts.setdefault(tag, []).append(tensor)
    out: Dict[str, np.ndarray] = {}
    for tag, chunks in parts.items():
        dims, ttype, _ = chunks[0]
        if ttype not in _TYPES:
            raise CheckpointFormatError(f"{tag}: unsupported tensor_type {ttype!r}")
        raw = b"".join(c[2] for c in chunks)
        arr = np.frombuffer(raw, dtype=_TYPES[ttype])
        shape = tuple(dims)
        if shape == (0,) or shape == ():  # MindSpore writes scalars with dims [0]
            shape = () if arr.size == 1 else (arr.size,)
        if int(np.prod(shape, dtype=np.int64)) != arr.size:
            raise CheckpointFormatError(f"{tag}: dims {dims} do not match {arr.size} elements of {ttype}")
        out[tag] = arr.reshape(shape).copy()
    return out


def _enc_varint(v: int) -> bytes:
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(num: int, payload: bytes) -> bytes:
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def save_checkpoint(params: Dict[str, np.ndarray], ckpt_file_name: str) -> None:
    """Write ``{name: array}`` in the same format (what ``mindspore.save_checkpoint(net, path)`` produces for ``net``)."""
    with open(ckpt_file_name, "wb") as f:
        for tag, arr in params.items():
            a = np.asarray(arr)  # (np.ascontiguousarray would turn a 0-d scalar into shape (1,))
            if a.dtype not in _NAMES:
                raise CheckpointFormatError(f"{tag}: dtype {a.dtype} has no MindSpore tensor_type")
            dims = a.shape if a.ndim else (0,)
            tensor = b"".join(_enc_varint((1 << 3) | 0) + _enc_varint(d) for d in dims)
            tensor += _ld(2, _NAMES[a.dtype].encode()) + _ld(3, np.ascontiguousarray(a).tobytes())
            value = _ld(1, tag.encode("utf-8")) + _ld(2, tensor)
            f.write(_ld(1, value))


_STRIP = ("network.", "net.", "_backbone.")  # cells that wrap Net in the reference: NetWithLoss.net, TrainOneStepCell.network


def load_param_into_net(net, parameter_dict: Dict[str, np.ndarray], strict_load: bool = False) -> List[str]:
    """``mindspore.load_param_into_net(net, parameter_dict)``: copy matching parameters / BatchNorm statistics into a
    network of this package and return the names of the network's parameters that were NOT loaded.

    Wrapper prefixes of the reference's training cells (``net.``, ``network.``) are stripped; optimizer state
    (``moment1.*``, ``moment2.*``, ``global_step``, ``learning_rate`` ...) is ignored.  A shape mismatch raises
    ``RuntimeError`` like MindSpore; ``strict_load`` additionally requires equal dtypes (fp16 checkpoints are cast to
    the network's fp32 otherwise)."""
    import torch
    own = dict(net.state_dict())
    lookup = {}
    for name, arr in parameter_dict.items():
        key = name
        changed = True
        while key not in own and changed:
            changed = False
            for p in _STRIP:
                if key.startswith(p):
                    key, changed = key[len(p):], True
        if key in own and key not in lookup:
            lookup[key] = arr
    missing = []
    with torch.no_grad():
        for name, dst in own.items():
            src = lookup.get(name)
            if src is None:
                if not name.endswith("num_batches_tracked"):
                    missing.append(name)
                continue
            src = np.asarray(src)
            if tuple(src.shape) != tuple(dst.shape):
                raise RuntimeError(f"For 'load_param_into_net', {name} in the argument 'net' should have the same shape as "
                                   f"{name} in the argument 'parameter_dict'. But got its shape {tuple(dst.shape)} in the "
                                   f"argument 'net' and shape {tuple(src.shape)} in the argument 'parameter_dict'.")
            t = torch.from_numpy(src.copy())
            if strict_load and t.dtype != dst.dtype:
                raise RuntimeError(f"{name}: dtype {t.dtype} in the checkpoint, {dst.dtype} in the network (strict_load)")
            dst.copy_(t.to(dst.dtype))
    if hasattr(net, "invalidate_plans"):
        net.invalidate_plans()  # packed weights / folded BatchNorm of recorded launch plans are stale now
    return missing
