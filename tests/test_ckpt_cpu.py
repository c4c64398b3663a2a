"""MindSpore .ckpt reader / load_param_into_net - host logic, runs on CPU.

The wire format is pinned by a fixture assembled byte by byte from the protobuf encoding rules and the
checkpoint.proto schema (independent of the package's own writer); PARITY UNPINNED against a real MindSpore file."""
import struct

import numpy as np
import pytest
import torch

import mindpose_amd as mp
from mindpose_amd.utils import ckpt


def _hand_fixture():
    content = struct.pack("<6f", 1.0, -2.0, 3.5, 0.0, 1e-3, 7.0)
    tensor = bytes([0x08, 0x02, 0x08, 0x03])                    # dims: 2, 3   (field 1, varint, unpacked - proto2)
    tensor += bytes([0x12, 0x07]) + b"Float32"                  # tensor_type  (field 2, length-delimited)
    tensor += bytes([0x1A, len(content)]) + content             # tensor_content (field 3)
    value = bytes([0x0A, 0x03]) + b"a.b" + bytes([0x12, len(tensor)]) + tensor
    return bytes([0x0A, len(value)]) + value, np.array([[1.0, -2.0, 3.5], [0.0, 1e-3, 7.0]], np.float32)


def test_reads_hand_assembled_checkpoint(tmp_path):
    raw, want = _hand_fixture()
    p = tmp_path / "hand.ckpt"
    p.write_bytes(raw)
    got = ckpt.load_checkpoint(str(p))
    assert list(got) == ["a.b"] and got["a.b"].dtype == np.float32 and np.array_equal(got["a.b"], want)
    # the package's writer produces exactly these bytes
    q = tmp_path / "own.ckpt"
    ckpt.save_checkpoint({"a.b": want}, str(q))
    assert q.read_bytes() == raw


def test_wire_format_variants(tmp_path):
    raw, want = _hand_fixture()
    # (i) two serialised Checkpoint messages back to back merge (MindSpore writes in slices)
    raw2 = raw.replace(b"a.b", b"c.d")
    p = tmp_path / "two.ckpt"
    p.write_bytes(raw + raw2)
    got = ckpt.load_checkpoint(str(p))
    assert list(got) == ["a.b", "c.d"] and np.array_equal(got["c.d"], want)
    # (ii) packed dims, an unknown extra field in Value, a tensor split over two Values with the same tag
    c1, c2 = want.tobytes()[:8], want.tobytes()[8:]

    def value(content, dims_packed):
        t = (bytes([0x0A, 0x02, 0x02, 0x03]) if dims_packed else bytes([0x08, 0x02, 0x08, 0x03]))
        t += bytes([0x12, 0x07]) + b"Float32" + bytes([0x1A, len(content)]) + content
        v = bytes([0x0A, 0x01]) + b"w" + bytes([0x12, len(t)]) + t + bytes([0x18, 0x05])  # field 3 varint: ignored
        return bytes([0x0A, len(v)]) + v
    p.write_bytes(value(c1, True) + value(c2, False))
    got = ckpt.load_checkpoint(str(p))
    assert np.array_equal(got["w"], want)
    # (iii) scalar written with dims [0] (global_step), Int32; filter_prefix drops optimizer state
    ckpt.save_checkpoint({"global_step": np.array(7, np.int32), "moment1.x": np.zeros(3, np.float32),
                          "x": np.ones(3, np.float16)}, str(p))
    got = ckpt.load_checkpoint(str(p), filter_prefix=["moment1."])
    assert list(got) == ["global_step", "x"] and got["global_step"].shape == () and int(got["global_step"]) == 7
    assert got["x"].dtype == np.float16
    # (iv) errors are loud
    p.write_bytes(raw[:-5])
    with pytest.raises(ckpt.CheckpointFormatError):
        ckpt.load_checkpoint(str(p))
    p.write_bytes(raw.replace(b"Float32", b"Complex"))
    with pytest.raises(ckpt.CheckpointFormatError):
        ckpt.load_checkpoint(str(p))
    p.write_bytes(b"")
    with pytest.raises(ckpt.CheckpointFormatError):
        ckpt.load_checkpoint(str(p))


def test_round_trip_into_network(tmp_path):
    src = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=3)
    params = {k: v.numpy() for k, v in src.state_dict().items()}
    assert len(params) > 1400 and "backbone.stage2.0.branches.0.0.bn1.moving_variance" in params
    # as a training checkpoint would hold them: wrapped in NetWithLoss (net.) + optimizer state
    saved = {"net." + k: v for k, v in params.items()}
    saved["moment1.net.backbone.conv1.weight"] = np.zeros_like(params["backbone.conv1.weight"])
    saved["global_step"] = np.array(100, np.int32)
    p = tmp_path / "hrnet.ckpt"
    ckpt.save_checkpoint(saved, str(p))
    loaded = ckpt.load_checkpoint(str(p))
    dst = mp.create_network("hrnet_w32", "hrnet_head")
    missing = ckpt.load_param_into_net(dst, loaded)
    assert missing == []
    for k, v in dst.state_dict().items():
        assert torch.equal(v, src.state_dict()[k]), k
    # partial checkpoints report what stayed untouched; shape mismatches raise like MindSpore
    part = {k: v for k, v in params.items() if k.startswith("backbone.")}
    missing = ckpt.load_param_into_net(mp.create_network("hrnet_w32", "hrnet_head"), part)
    assert sorted(missing) == ["head.head.bias", "head.head.weight"]
    bad = dict(params)
    bad["head.head.weight"] = np.zeros((17, 48, 1, 1), np.float32)
    with pytest.raises(RuntimeError, match="should have the same shape"):
        ckpt.load_param_into_net(mp.create_network("hrnet_w32", "hrnet_head"), bad)
    # fp16 checkpoints are cast to the network's fp32 unless strict_load
    half = {k: v.astype(np.float16) if v.dtype == np.float32 else v for k, v in params.items()}
    net16 = mp.create_network("hrnet_w32", "hrnet_head")
    assert ckpt.load_param_into_net(net16, half) == []
    assert net16.state_dict()["backbone.conv1.weight"].dtype == torch.float32
    with pytest.raises(RuntimeError, match="strict_load"):
        ckpt.load_param_into_net(mp.create_network("hrnet_w32", "hrnet_head"), half, strict_load=True)
