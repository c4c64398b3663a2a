"""BASELINE configs[3] at its stated size on one GPU: the HRNet-W32 256x192 training step as bench.py runs it - amp O2,
captured into a hipGraph with the HRModule branches on side streams - plus the oracle yardsticks of a full training step.

  * graph replay (branch side streams, fork / join as graph dependencies) == eager single-stream step, bit for bit, at
    256x192 with a batch large enough for the deep-branch launches to overlap (N = 48; MINDPOSE_TEST_FULL_BATCH=128 for the
    reference's per-device batch): an ordering bug in the stream fork / join would show here, not at 64x64.
  * fp32 training forward at 256x192 (N = 8): loss equal to the oracle's (batch-statistics BatchNorm) to 1e-6.
  * amp O2 full step against the ORACLE's amp-O2 training emulation (oracle/nets.py net_forward_train(amp=True)), not against
    this repository's own fp32 path.
  * evaluation between graphed steps sees the updated weights (recorded inference plans are dropped by the graphed step).
Reference: tools/train.py:170-233 (Model(amp_level="O2", loss_scale_manager=...).train).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import mindpose_amd as mp  # noqa: E402
from oracle import nets as onets  # noqa: E402

DEV = torch.device("cuda:0")


def _data(n, h, w, seed=11):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 3, h, w, generator=g)
    kp = torch.empty(n, 17, 3)
    kp[..., 0] = torch.rand(n, 17, generator=g) * (w + 40) - 20
    kp[..., 1] = torch.rand(n, 17, generator=g) * (h + 40) - 20
    kp[..., 2] = (torch.rand(n, 17, generator=g) < 0.7).float()
    return x, kp


def _build(amp):
    from mindpose_amd.utils import AdamWeightDecay
    torch.manual_seed(0)
    net = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=0).to(DEV).train()
    if amp:
        mp.models.auto_mixed_precision(net, "O2")
    nwl = mp.create_network_with_loss(net, mp.create_loss("joint_mse", use_target_weight=True), has_extra_inputs=True)
    opt = AdamWeightDecay(net, lr=1e-3, weight_decay=0.05, filter_bias_and_bn=True, overlap=False)
    return net, nwl, opt


def test_config4_step_at_full_size_graph_with_branch_streams_equals_eager():
    from mindpose_amd.utils import DynamicLossScaleManager, GraphedTrainStep
    n = int(os.environ.get("MINDPOSE_TEST_FULL_BATCH", "48"))
    x, kp = _data(n, 256, 192)
    x, kp = x.to(DEV), kp.to(DEV)
    tgt = mp.TopDownGenerateTarget(config=dict(image_size=[192, 256], heatmap_size=[48, 64]), sigma=2.0)
    target, weight = tgt(kp)
    scale = 1024.0
    # eager, everything on one stream
    net_e, nwl_e, opt_e = _build(True)
    opt_e.zero_grad()
    loss_e = nwl_e(x, target, weight)
    (loss_e * scale).backward()
    torch.cuda.synchronize()
    arena_e = opt_e.grads.arena.clone()
    assert torch.isfinite(arena_e).all() and float(arena_e.abs().max()) > 0
    # the captured step: HRModule branches on side streams inside the graph
    net_g, nwl_g, opt_g = _build(True)
    mgr = DynamicLossScaleManager(init_loss_scale=scale)
    step = GraphedTrainStep(nwl_g, opt_g, (x, target, weight), loss_scale_manager=mgr, warmup=2)
    opt_g.grads.rearm()
    step.graph.replay()
    torch.cuda.synchronize()
    assert float(step.static_loss) == float(loss_e.detach())
    assert torch.equal(opt_g.grads.arena, arena_e), "graph replay with branch side streams differs from the eager single-stream step"
    # replay again: the graph zeroes and refills the arena (bit-reproducible run to run)
    step.graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(opt_g.grads.arena, arena_e)
    # and a real step updates
    before = opt_g.flat.clone()
    step(x, target, weight)
    assert step.updated and not torch.equal(opt_g.flat, before)


def test_fp32_training_forward_loss_at_256x192_vs_oracle():
    """Loss of the fp32 training forward (batch statistics) on an 8-crop slice at the recipe's resolution == the oracle's, 1e-6."""
    x, kp = _data(8, 256, 192, seed=5)
    net, nwl, _ = _build(False)
    tgt = mp.TopDownGenerateTarget(config=dict(image_size=[192, 256], heatmap_size=[48, 64]), sigma=2.0)
    target, weight = tgt(kp.to(DEV))
    params = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    loss = nwl(x.to(DEV), target, weight)
    with torch.no_grad():
        out = onets.net_forward_train(params, x, "hrnet_w32", "hrnet_head")
        ref = float((((out - target.cpu()) ** 2) * weight.cpu()[..., None, None]).mean())
    assert abs(float(loss.detach()) - ref) <= 1e-6 * abs(ref), (float(loss.detach()), ref)


def test_o2_training_step_vs_oracle_amp_emulation():
    """Loss and every parameter gradient of the amp-O2 HIP step against the oracle's amp-O2 emulation (fp16 operands and cell
    outputs, fp32 accumulation / statistics) - and, as the yardstick for how far two fp16 roundings of the same graph may sit
    apart, against the oracle's fp32 gradients: the HIP path must be at least as close to fp32 as the emulation is (x1.5)."""
    net = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=0)
    cpu_state = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}  # before the network moves to the GPU

    def leaf_params():
        d = {k: v.clone() for k, v in cpu_state.items()}
        for k, v in d.items():
            if v.dtype.is_floating_point and not k.endswith(("moving_mean", "moving_variance")):
                v.requires_grad_()
        return d

    x, kp = _data(4, 128, 96)
    net = net.to(DEV).train()
    mp.models.auto_mixed_precision(net, "O2")
    tgt = mp.TopDownGenerateTarget(config=dict(image_size=[96, 128], heatmap_size=[24, 32]), sigma=2.0)
    target, weight = tgt(kp.to(DEV))
    nwl = mp.create_network_with_loss(net, mp.create_loss("joint_mse", use_target_weight=True), has_extra_inputs=True)
    scale = 1024.0
    loss = nwl(x.to(DEV), target, weight)
    (loss * scale).backward()
    got = {k: (v.grad / scale).double().cpu().flatten() for k, v in net.named_parameters()}

    def oracle(amp):
        p = leaf_params()
        out = onets.net_forward_train(p, x, "hrnet_w32", "hrnet_head", amp=amp)
        l = (((out - target.cpu()) ** 2) * weight.cpu()[..., None, None]).mean()
        (l * scale).backward()
        return float(l.detach()), {k: (p[k].grad / scale).double().flatten() for k in got}

    l_amp, g_amp = oracle(True)
    l_f32, g_f32 = oracle(False)
    assert abs(float(loss.detach()) - l_amp) <= 5e-3 * abs(l_amp), (float(loss.detach()), l_amp, l_f32)

    def cos(a, b):
        return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-300))

    c_amp = {k: cos(got[k], g_amp[k]) for k in got}          # HIP O2 vs the emulation
    c_hip_f32 = {k: cos(got[k], g_f32[k]) for k in got}      # HIP O2 vs fp32 truth
    c_emu_f32 = {k: cos(g_amp[k], g_f32[k]) for k in got}    # emulation vs fp32 truth
    all_hip, all_amp, all_f32 = (torch.cat([d[k] for k in got]) for d in (got, g_amp, g_f32))
    print(f"O2 step: loss {float(loss.detach()):.6f} / emulation {l_amp:.6f} / fp32 {l_f32:.6f}; global cosine vs emulation "
          f"{cos(all_hip, all_amp):.5f}, vs fp32 {cos(all_hip, all_f32):.5f} (emulation vs fp32 {cos(all_amp, all_f32):.5f}); "
          f"per-tensor vs emulation median {np.median(list(c_amp.values())):.5f} min {min(c_amp.values()):.4f}")
    assert cos(all_hip, all_amp) > 0.98 and np.median(list(c_amp.values())) > 0.99 and min(c_amp.values()) > 0.9
    # per-tensor floor 0.97 (measured minimum 0.9736: BatchNorm parameters of the deepest layers, whose gradient is a
    # small difference of large sums - the emulation itself sits that far from the fp32 gradient on them)
    low = sorted((k for k in got if c_amp[k] < 0.98), key=lambda k: c_amp[k])
    for k in low[:8]:
        print(f"  lowest: {k} numel {got[k].numel()} cos(hip, emu) {c_amp[k]:.4f} cos(hip, f32) {c_hip_f32[k]:.4f} "
              f"cos(emu, f32) {c_emu_f32[k]:.4f}")
    assert min(c_amp.values()) > 0.97
    # distance to the fp32 gradients: not worse than the op-by-op emulation of the reference's recipe
    assert 1 - cos(all_hip, all_f32) <= 1.5 * (1 - cos(all_amp, all_f32)) + 1e-4
    assert np.median([1 - c for c in c_hip_f32.values()]) <= 1.5 * np.median([1 - c for c in c_emu_f32.values()]) + 1e-4
    # one layer from the loss: tight against the emulation
    for name in ("head.head.weight", "head.head.bias"):
        rel = float((got[name] - g_amp[name]).abs().max() / g_amp[name].abs().max())
        assert rel < 5e-3, (name, rel)


def test_eval_between_graphed_steps_sees_updated_weights():
    """recorded inference plans (packed weights, folded BatchNorm) are dropped by every graphed step that updates the
    parameters, so the reference's train-with-per-epoch-eval loop evaluates the CURRENT weights."""
    from mindpose_amd.utils import GraphedTrainStep
    x, kp = _data(4, 64, 64, seed=3)
    x, kp = x.to(DEV), kp.to(DEV)
    tgt = mp.TopDownGenerateTarget(config=dict(image_size=[64, 64], heatmap_size=[16, 16]), sigma=2.0)
    target, weight = tgt(kp)
    net, nwl, opt = _build(False)
    step = GraphedTrainStep(nwl, opt, (x, target, weight), warmup=2)

    def evaluate():
        net.eval()
        out = net(x).clone()
        net.train()
        return out

    h0 = evaluate()
    assert len(net._plans) == 1
    for _ in range(3):
        step(x, target, weight)
    assert len(net._plans) == 0  # dropped by the step
    h1 = evaluate()
    assert not torch.equal(h0, h1)
    # a freshly built network holding the same parameters / statistics gives the same evaluation
    fresh = mp.create_network("hrnet_w32", "hrnet_head")
    fresh.load_state_dict({k: v.detach().cpu().clone() for k, v in net.state_dict().items()})
    fresh = fresh.to(DEV).eval()
    assert torch.equal(fresh(x), h1)
    for _ in range(2):
        step(x, target, weight)
    h2 = evaluate()
    assert not torch.equal(h1, h2)
