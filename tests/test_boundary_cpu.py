"""CPU-side checks of the drop-in boundary: registry names, factory signatures, error behaviour,
parameter names / counts, and that the C-ABI library loads and exports every declared symbol."""
import inspect
import os
import re
import subprocess

import pytest
import torch

import mindpose_amd as mp
from mindpose_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session", autouse=True)
def built_library():
    subprocess.run(["make", "-C", os.path.join(ROOT, "mindpose_amd", "csrc"), "-j", "8"], check=True,
                   stdout=subprocess.DEVNULL)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mindpose_hip.h")).read()
    declared = set(re.findall(r"\b(mp_[a-z0-9_]+)\s*\(", header))
    declared -= {"mp_conv_desc"}
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.mp_version().decode().startswith("mindpose_hip")
    assert lib.mp_error_string(-2).decode() == "bad shape"
    # argument validation happens before any HIP call, so it is testable without a GPU
    assert lib.mp_decode_topdown(None, None, None, None, None, None, None, 1, 17, 64, 48, 0, 0, 1, 200.0, None, 11, None) == -1
    assert lib.mp_joints_mse_workspace_bytes(128, 17) >= 128 * 17 * 4
    assert lib.mp_conv_packed_weight_bytes(17, 3, 3, 3) == 4 * 9 * 32 * 4
    assert lib.mp_conv2d_fwd(None, None, None, None, None, None, None, None, None) == -1
    assert lib.mp_plan_size(None) == -1


def test_registry_names_resolve():
    names = {
        "backbone": ["HRNet", "hrnet_w32", "hrnet_w48", "ResNet", "resnet50", "resnet101", "resnet152"],
        "head": ["HRNetHead", "hrnet_head", "SimpleBaselineHead", "simple_baseline_head"],
        "decoder": ["TopDownHeatMapDecoder", "topdown_heatmap"],
        "loss": ["JointsMSELoss", "joint_mse"],
        "transform": ["TopDownGenerateTarget", "topdown_generate_target", "TopDownBoxToCenterScale",
                      "topdown_box_to_center_scale", "TopDownAffine", "topdown_affine", "topdown_horizontal_random_flip",
                      "topdown_halfbody_transform", "topdown_randomscale_rotation"],
        "inferencer": ["TopDownHeatMapInferencer", "topdown_heatmap"],
    }
    for module, comps in names.items():
        for c in comps:
            assert callable(mp.entrypoint(module, c))
    with pytest.raises(ValueError, match="Unkown module"):
        mp.entrypoint("nope", "x")
    with pytest.raises(ValueError, match="Unkown components"):
        mp.entrypoint("backbone", "vgg")


def test_factory_signatures_match_reference():
    sig = inspect.signature(mp.create_network)
    assert list(sig.parameters) == ["backbone_name", "head_name", "neck_name", "backbone_pretrained",
                                    "backbone_ckpt_url", "in_channels", "neck_out_channels", "num_joints",
                                    "backbone_args", "neck_args", "head_args"]
    assert list(inspect.signature(mp.create_backbone).parameters) == ["name", "pretrained", "ckpt_url", "in_channels", "kwargs"]
    assert list(inspect.signature(mp.create_head).parameters) == ["name", "in_channels", "num_joints", "kwargs"]
    assert list(inspect.signature(mp.create_eval_network).parameters) == ["net", "decoder", "output_raw"]
    assert list(inspect.signature(mp.create_network_with_loss).parameters) == ["net", "loss", "has_extra_inputs"]
    dsig = inspect.signature(mp.entrypoint("decoder", "topdown_heatmap").__init__)
    assert [(k, v.default) for k, v in list(dsig.parameters.items())[1:]] == [
        ("pixel_std", 200.0), ("to_original", True), ("shift_coordinate", False), ("use_udp", False),
        ("dark_udp_refine", False), ("kernel_size", 11)]


@pytest.mark.parametrize("backbone,head,published_m", [
    ("hrnet_w32", "hrnet_head", 28.59), ("hrnet_w48", "hrnet_head", 63.68), ("resnet50", "simple_baseline_head", 34.05),
    ("resnet101", "simple_baseline_head", 53.10), ("resnet152", "simple_baseline_head", 68.79)])
def test_param_counts_match_published(backbone, head, published_m):
    # configs/hrnet/README.md:17-18, configs/simple_baseline/README.md:17-19 (BN moving stats included)
    net = mp.create_network(backbone, head)
    count = sum(t.numel() for t in list(net.parameters()) + list(net.buffers()))
    assert round(count / 1e6, 2) == published_m


def test_parameter_names_follow_reference_cells():
    sd = mp.create_network("hrnet_w32", "hrnet_head").state_dict()
    for key in ["backbone.conv1.weight", "backbone.bn1.gamma", "backbone.bn1.moving_variance",
                "backbone.layer1.0.down_sample.0.weight", "backbone.layer1.0.down_sample.1.beta",
                "backbone.transition1.0.0.weight", "backbone.transition1.1.0.0.weight",
                "backbone.stage2.0.branches.0.0.conv1.weight", "backbone.stage2.0.fuse_layers.0.1.0.weight",
                "backbone.stage2.0.fuse_layers.1.0.0.0.weight", "backbone.stage2.0.fuse_layers.1.0.0.1.moving_mean",
                "backbone.transition2.2.0.0.weight", "backbone.stage4.2.fuse_layers.0.3.1.gamma",
                "head.head.weight", "head.head.bias"]:
        assert key in sd, key
    assert "backbone.stage4.2.fuse_layers.1.0.0.0.weight" not in sd  # last module: only fuse row 0
    sd = mp.create_network("resnet50", "simple_baseline_head").state_dict()
    for key in ["backbone.conv1.weight", "backbone.layer2.0.down_sample.0.weight", "head.deconv_layer.0.weight",
                "head.deconv_layer.1.gamma", "head.deconv_layer.6.weight", "head.final_layer.bias"]:
        assert key in sd, key
    assert tuple(sd["head.deconv_layer.0.weight"].shape) == (2048, 256, 4, 4)


def test_errors_without_gpu_are_loud():
    with pytest.raises(ValueError):
        mp.create_decoder("topdown_heatmap", shift_coordinate=True, dark_udp_refine=True)
    net = mp.create_network("hrnet_w32", "hrnet_head")
    with pytest.raises(_lib.MindposeHipError):
        net(torch.zeros(1, 3, 64, 48))  # CPU tensor: the HIP path has no CPU fallback
    with pytest.raises(_lib.MindposeHipError):
        mp.create_loss("joint_mse")(torch.zeros(1, 1, 2, 2), torch.zeros(1, 1, 2, 2))
    with pytest.raises(ValueError):
        mp.TopDownHeatMapInferencer(None, config=dict(has_heatmap_output=True, hflip_tta=True, shift_heatmap=False,
                                                      flip_pairs=[[1, 2]]))
    with pytest.raises(FileNotFoundError):
        mp.create_backbone("hrnet_w32", pretrained=True, ckpt_url="/nonexistent.ckpt")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "mindpose_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, os.path.join(dirpath, f)
