"""The amp-O2 training step EXACTLY as bench.py::train_bench builds it - tuner on, BatchNorm fusion on, grouped weight gradients,
exchange-unit rows on side streams, forward + loss + backward captured into a hipGraph - at the recipe's 256x192 resolution, against
the ORACLE (oracle/nets.py net_forward_train: the amp-O2 emulation and the fp32 graph), not against another HIP path.

BatchNorm batch statistics forbid slicing a batch, so the whole batch goes through the CPU oracle: N = 32 by default (~2 minutes of
host time), MINDPOSE_TEST_BENCH_BATCH=128 for the reference's per-device batch.  Asserts = tests/test_gpu_train_full.py's
128x96 yardsticks: loss within 5e-3 of the emulation, global gradient cosine > 0.98, per-tensor floor 0.97, and the HIP step at
least as close to the fp32 gradients as the emulation is (x 1.5).  Reference: tools/train.py:170-233.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import mindpose_amd as mp  # noqa: E402
from oracle import nets as onets  # noqa: E402

DEV = torch.device("cuda:0")


def test_graphed_o2_step_at_bench_shape_vs_oracle():
    from mindpose_amd.utils import AdamWeightDecay, DynamicLossScaleManager, GraphedTrainStep
    n = int(os.environ.get("MINDPOSE_TEST_BENCH_BATCH", "32"))
    scale = 1024.0
    # bench.py::train_bench, line by line (data: its seed-1000 generator)
    net = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=0)
    cpu_state = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    net = net.to(DEV).train()
    mp.models.auto_mixed_precision(net, "O2")
    scaler = DynamicLossScaleManager(init_loss_scale=scale)
    nwl = mp.create_network_with_loss(net, mp.create_loss("joint_mse", use_target_weight=True), has_extra_inputs=True)
    opt = AdamWeightDecay(net, lr=1e-3, weight_decay=0.05, filter_bias_and_bn=True, overlap=False)
    tgt = mp.TopDownGenerateTarget(config=dict(image_size=[192, 256], heatmap_size=[48, 64]), sigma=2.0)
    gen = torch.Generator(device="cpu").manual_seed(1000)
    image = torch.randn(n, 3, 256, 192, generator=gen)
    kp = torch.empty(n, 17, 3)
    kp[..., 0] = torch.rand(n, 17, generator=gen) * 232 - 20
    kp[..., 1] = torch.rand(n, 17, generator=gen) * 296 - 20
    kp[..., 2] = (torch.rand(n, 17, generator=gen) < 0.7).float()
    target, weight = tgt(kp.to(DEV))
    gstep = GraphedTrainStep(nwl, opt, (image.to(DEV), target, weight), loss_scale_manager=scaler)
    # one replay of the captured forward + loss + backward (no update): the gradient arena holds scale x gradient
    opt.grads.rearm()
    gstep.graph.replay()
    torch.cuda.synchronize()
    assert scaler.loss_scale == scale, "the warm-up steps overflowed: the comparison below would use another scale"
    loss = float(gstep.static_loss.detach())
    got = {k: (v.grad / scale).double().cpu().flatten() for k, v in net.named_parameters()}
    assert all(torch.isfinite(g).all() for g in got.values())

    def oracle(amp):
        p = {k: v.clone() for k, v in cpu_state.items()}
        for k, v in p.items():
            if v.dtype.is_floating_point and not k.endswith(("moving_mean", "moving_variance")):
                v.requires_grad_()
        out = onets.net_forward_train(p, image, "hrnet_w32", "hrnet_head", amp=amp)
        l = (((out - target.cpu()) ** 2) * weight.cpu()[..., None, None]).mean()
        (l * scale).backward()
        return float(l.detach()), {k: (p[k].grad / scale).double().flatten() for k in got}

    l_amp, g_amp = oracle(True)
    l_f32, g_f32 = oracle(False)

    def cos(a, b):
        return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-300))

    c_amp = {k: cos(got[k], g_amp[k]) for k in got}
    c_hip_f32 = {k: cos(got[k], g_f32[k]) for k in got}
    c_emu_f32 = {k: cos(g_amp[k], g_f32[k]) for k in got}
    all_hip, all_amp, all_f32 = (torch.cat([d[k] for k in got]) for d in (got, g_amp, g_f32))
    print(f"graphed O2 step at 256x192, N={n}: loss {loss:.6f} / emulation {l_amp:.6f} / fp32 {l_f32:.6f}; global cosine vs emulation "
          f"{cos(all_hip, all_amp):.5f}, vs fp32 {cos(all_hip, all_f32):.5f} (emulation vs fp32 {cos(all_amp, all_f32):.5f}); per-tensor "
          f"vs emulation median {np.median(list(c_amp.values())):.5f} min {min(c_amp.values()):.4f}")
    assert abs(loss - l_amp) <= 5e-3 * abs(l_amp), (loss, l_amp, l_f32)
    assert cos(all_hip, all_amp) > 0.98 and np.median(list(c_amp.values())) > 0.99
    for k in sorted((k for k in got if c_amp[k] < 0.98), key=lambda k: c_amp[k])[:8]:
        print(f"  lowest: {k} numel {got[k].numel()} cos(hip, emu) {c_amp[k]:.4f} cos(hip, f32) {c_hip_f32[k]:.4f} "
              f"cos(emu, f32) {c_emu_f32[k]:.4f}")
    assert min(c_amp.values()) > 0.97
    assert 1 - cos(all_hip, all_f32) <= 1.5 * (1 - cos(all_amp, all_f32)) + 1e-4
    assert np.median([1 - c for c in c_hip_f32.values()]) <= 1.5 * np.median([1 - c for c in c_emu_f32.values()]) + 1e-4
    opt.close()
