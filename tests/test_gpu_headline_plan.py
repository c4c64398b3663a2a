"""The plan the benchmark times, held to the oracle (VERDICT r2 "what's weak" 2 / 3).

`bench.py` runs N = 128 with the per-shape tuner on and four execution lanes; the oracle comparisons of test_gpu_conv.py run at
N <= 3, where the tuner picks other forms (one-team Winograd, other GEMM tiles).  Outputs are per-image independent, so the
N = 128 plan is built exactly as bench.py builds it and images [0:4] + [124:128] are compared with `oracle/nets.py`
(1e-3 of the heat-map scale + arg-max equality wherever the oracle's top-1 / top-2 margin exceeds that tolerance).

Reproducibility: the tuner chooses between numerically different algorithms (Winograd vs direct, one GEMM launch vs four phase
convs) from timings, so heat-map BITS depend on its choices.  `MINDPOSE_TUNE_CACHE=<file>` pins them: a second, fresh process
replaying the same file must produce bit-identical heat maps.
"""
import hashlib
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

import mindpose_amd as mp  # noqa: E402
from oracle import nets as onets  # noqa: E402

DEV = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 128
SLICE = list(range(0, 4)) + list(range(N - 4, N))


def _bench_like_batch(n, h, w, seed=1000):
    """bench.py's synthetic crops: randn from a CPU generator seeded 1000 + rank."""
    return torch.randn(n, 3, h, w, generator=torch.Generator(device="cpu").manual_seed(seed))


def _check_slice(got, ref, tol):
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < tol, f"normalised max error {err}"
    n, k = got.shape[:2]
    rf = ref.reshape(n, k, -1)
    top2 = rf.topk(2, dim=2).values
    safe = (top2[..., 0] - top2[..., 1]) > 2.0 * tol * ref.abs().max()
    assert safe.float().mean() > 0.5
    assert torch.equal(got.reshape(n, k, -1).argmax(2)[safe], rf.argmax(2)[safe])
    return err


@pytest.mark.parametrize("backbone,head", [("hrnet_w32", "hrnet_head"), ("resnet50", "simple_baseline_head")])
def test_headline_plan_n128_fp32_vs_oracle(backbone, head):
    net = mp.init_synthetic(mp.create_network(backbone, head), seed=0).to(DEV).eval()
    x = _bench_like_batch(N, 256, 192)
    image = net.input_buffer((N, 3, 256, 192), DEV)  # as bench.py: crops written straight into the plan's input buffer
    image.copy_(x)
    got = net(image)[SLICE].cpu()
    plan = net.get_plan((N, 3, 256, 192), DEV)
    kinds = {e["kind"] for e in plan.layer_info}
    assert "conv_winograd" in kinds  # the tuned plan of the benchmark, not the small-batch forms
    if backbone == "hrnet_w32":
        assert "barrier" in kinds  # four execution lanes
    params = {k: v.cpu() for k, v in net.state_dict().items()}
    ref = onets.net_forward(params, x[SLICE], backbone, head)
    _check_slice(got, ref, 1e-3)


@pytest.mark.parametrize("n", [1, 32])
def test_small_batch_plans_fp32_vs_oracle(n):
    """SURVEY 8(d) names N in {1, 32, 128, 256}: the plans `bench.py --sweep` times at N = 1 and N = 32 (their own tuner choices - the
    K-split small-problem kernel on the deep branches, the stage-1 chain launches on few tiles) against `oracle/nets.py`."""
    net = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=0).to(DEV).eval()
    x = _bench_like_batch(n, 256, 192)
    image = net.input_buffer((n, 3, 256, 192), DEV)
    image.copy_(x)
    sl = sorted(set(list(range(0, min(2, n))) + list(range(max(0, n - 2), n))))
    got = net(image)[sl].cpu()
    plan = net.get_plan((n, 3, 256, 192), DEV)
    variants = [plan.entry_info(i).get("variant") for i in range(len(plan)) if plan.layer_info[i]["kind"] == "conv"]
    assert any(v in (11, 12) for v in variants), "the small-problem kernel is in neither plan"
    assert [e["kind"] for e in plan.layer_info].count("pwchain_f32") == 4
    params = {k: v.cpu() for k, v in net.state_dict().items()}
    ref = onets.net_forward(params, x[sl], "hrnet_w32", "hrnet_head")
    _check_slice(got, ref, 1e-3)


@pytest.mark.parametrize("n", [1, 32])
def test_small_batch_plans_amp_o2_vs_amp_oracle(n):
    """The amp-O2 plans of the same two batch sizes (fused-block band heights and conv variants tuned per N) against the oracle's fp32
    graph and its op-by-op amp emulation - the asserts of the N = 128 test."""
    net = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=0).to(DEV).eval()
    mp.models.auto_mixed_precision(net, "O2")
    x = _bench_like_batch(n, 256, 192)
    image = net.input_buffer((n, 3, 256, 192), DEV)
    image.copy_(x)
    sl = sorted(set(list(range(0, min(2, n))) + list(range(max(0, n - 2), n))))
    got = net(image)[sl].cpu()
    params = {k: v.cpu() for k, v in net.state_dict().items()}
    ref32 = onets.net_forward(params, x[sl], "hrnet_w32", "hrnet_head")
    ref16 = onets.net_forward(params, x[sl], "hrnet_w32", "hrnet_head", amp=True)
    e_hip = float((got - ref32).abs().max() / ref32.abs().max())
    e_emul = float((ref16 - ref32).abs().max() / ref32.abs().max())
    assert e_hip <= 1.5 * e_emul + 1e-3, f"HIP fp16 {e_hip} vs op-by-op amp-O2 emulation {e_emul}"
    assert e_hip < 2e-2


def test_headline_plan_n128_amp_o2_vs_amp_oracle():
    net = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=0).to(DEV).eval()
    mp.models.auto_mixed_precision(net, "O2")
    x = _bench_like_batch(N, 256, 192)
    image = net.input_buffer((N, 3, 256, 192), DEV)
    image.copy_(x)
    got = net(image)[SLICE].cpu()
    params = {k: v.cpu() for k, v in net.state_dict().items()}
    ref32 = onets.net_forward(params, x[SLICE], "hrnet_w32", "hrnet_head")
    ref16 = onets.net_forward(params, x[SLICE], "hrnet_w32", "hrnet_head", amp=True)
    e_hip = float((got - ref32).abs().max() / ref32.abs().max())
    e_emul = float((ref16 - ref32).abs().max() / ref32.abs().max())
    # one rounding per stored tensor here, one per cell in the reference's amp O2: not further from fp32 than the emulation (x1.5)
    assert e_hip <= 1.5 * e_emul + 1e-3, f"HIP fp16 {e_hip} vs op-by-op amp-O2 emulation {e_emul}"
    assert e_hip < 2e-2
    n, k = got.shape[:2]
    rf = ref32.reshape(n, k, -1)
    top2 = rf.topk(2, dim=2).values
    safe = (top2[..., 0] - top2[..., 1]) > 2.5 * e_hip * ref32.abs().max()
    assert safe.float().mean() > 0.5
    assert torch.equal(got.reshape(n, k, -1).argmax(2)[safe], rf.argmax(2)[safe])


def test_config5_plan_2n128_amp_o2_vs_oracle():
    """BASELINE.json configs[4] at the size bench.py times it: HRNet-W48 384x288, N = 64 crops, amp O2, the flip test as ONE forward
    of 2N = 128 = [crops | mirrors] (PlannedModule.forward_flip_pair), UDP + DARK (k = 17) decode of the flip-averaged maps.  The
    plan is built as bench.py builds it (tuner on: weight-stationary / weights-in-registers picks, the stage-1 chain / dual launches,
    the fused first conv at 96x72); crops [0:2] + [62:64] AND their mirrors are compared with oracle/nets.py (fp32 and the amp-O2
    emulation), the decoded key points of those crops with oracle/decoder.py on the same heat maps."""
    import numpy as np
    from oracle import decoder as od
    from mindpose_amd.engine.inferencer.topdown_inferencer import COCO_FLIP_INDEX, _MultiRunNet
    n, h, w = 64, 384, 288
    sl = [0, 1, n - 2, n - 1]
    net = mp.init_synthetic(mp.create_network("hrnet_w48", "hrnet_head"), seed=0).to(DEV).eval()
    mp.models.auto_mixed_precision(net, "O2")
    dec = mp.create_decoder("topdown_heatmap", use_udp=True, dark_udp_refine=True, kernel_size=17).to(DEV)
    ev = mp.create_eval_network(net, dec, output_raw=True)
    mr = _MultiRunNet(ev, dec, np.array(COCO_FLIP_INDEX), shift_heatmap=False).to(DEV)
    gen = torch.Generator(device="cpu").manual_seed(1000)
    x = torch.randn(n, 3, h, w, generator=gen)
    image = net.input_buffer((2 * n, 3, h, w), DEV)[:n]  # as bench.py: the crops' half of the 2N plan's input
    image.copy_(x)
    center = torch.rand(n, 2, generator=gen) * 400
    scale = torch.rand(n, 2, generator=gen) * 2.7 + 0.3
    score = torch.rand(n, generator=gen)
    preds, boxes = mr(image, center.to(DEV), scale.to(DEV), score.to(DEV))
    plan = net.get_plan((2 * n, 3, h, w), DEV)
    assert plan.output.shape[0] == 2 * n  # the batched flip test ran (one forward of 128)
    both = plan.output.cpu().clone()
    got = torch.cat([both[sl], both[[n + i for i in sl]]])  # 4 crops, then their 4 mirrors
    xs = torch.cat([x[sl], torch.flip(x[sl], dims=[3])])
    params = {k: v.cpu() for k, v in net.state_dict().items()}
    ref32 = onets.net_forward(params, xs, "hrnet_w48", "hrnet_head")
    ref16 = onets.net_forward(params, xs, "hrnet_w48", "hrnet_head", amp=True)
    e_hip = float((got - ref32).abs().max() / ref32.abs().max())
    e_emul = float((ref16 - ref32).abs().max() / ref32.abs().max())
    assert e_hip <= 1.5 * e_emul + 1e-3, f"HIP fp16 {e_hip} vs op-by-op amp-O2 emulation {e_emul}"
    assert e_hip < 2e-2
    k = got.shape[1]
    rf = ref32.reshape(8, k, -1)
    top2 = rf.topk(2, dim=2).values
    safe = (top2[..., 0] - top2[..., 1]) > 2.5 * e_hip * ref32.abs().max()
    assert safe.float().mean() > 0.5
    assert torch.equal(got.reshape(8, k, -1).argmax(2)[safe], rf.argmax(2)[safe])
    # decode: the oracle's flip aggregation + UDP / DARK refinement on the HIP heat maps of the same crops
    avg = od.flip_aggregate(both[sl].numpy(), both[[n + i for i in sl]].numpy(), COCO_FLIP_INDEX, shift_heatmap=False)
    rp, rb, _ = od.decode(avg, center[sl].numpy(), scale[sl].numpy(), score[sl].numpy(), use_udp=True, dark_udp_refine=True, kernel_size=17)
    assert np.array_equal(boxes.cpu().numpy()[sl], rb)
    d = np.abs(preds.cpu().numpy()[sl][..., :2] - rp[..., :2])
    assert np.mean(d < 1e-2) > 0.98  # DARK's Hessian solve is ill-conditioned on a few flat maps (tests/test_gpu_pose_ops.py)


_CHILD = r"""
import hashlib, json, sys, torch
sys.path.insert(0, {root!r})
import mindpose_amd as mp
from mindpose_amd.models import layers
dev = torch.device("cuda:0")
net = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=0).to(dev).eval()
x = torch.randn(16, 3, 256, 192, generator=torch.Generator().manual_seed(7)).to(dev)
hm = net(x).cpu().contiguous()
plan = net.get_plan(x.shape, dev)
print(json.dumps(dict(sha=hashlib.sha256(hm.numpy().tobytes()).hexdigest(),
                      kinds=[e["kind"] for e in plan.layer_info],
                      tuned=len([k for k in layers._TUNE_CACHE if isinstance(k, str)]))))
"""


def _run_child(env):
    proc = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT)], env=env, stdout=subprocess.PIPE, text=True, timeout=600)
    assert proc.returncode == 0
    return json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])


def test_persisted_tune_cache_reproduces_heatmap_bits(tmp_path):
    """Two fresh processes sharing one MINDPOSE_TUNE_CACHE file: the first tunes and writes it, the second replays it without a
    trial launch; same plan forms, heat maps equal bit for bit."""
    cache = tmp_path / "tune.json"
    env = dict(os.environ, MINDPOSE_TUNE_CACHE=str(cache))
    first = _run_child(env)
    assert cache.exists()
    doc = json.loads(cache.read_text())
    assert doc["choices"] and "stamp" in doc
    second = _run_child(env)
    assert second["kinds"] == first["kinds"]
    assert second["sha"] == first["sha"], "heat maps differ between two processes replaying one tune cache"
    # and the switch for bit-reproducible runs WITHOUT a cache: no timing decides anything
    env0 = dict(os.environ, MINDPOSE_AUTOTUNE="0")
    env0.pop("MINDPOSE_TUNE_CACHE", None)
    a, b = _run_child(env0), _run_child(env0)
    assert a["sha"] == b["sha"] and a["tuned"] == 0
