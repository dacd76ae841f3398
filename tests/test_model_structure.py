"""CPU (-m "not gpu"): the drop-in surface of musev_amd.models -- state_dict keys / shapes, constructor
error conventions, registry names, loader flavours."""
import pytest
import torch

from oracle import unet3d

NARROW = dict(block_out_channels=(64, 128, 256, 256))


@pytest.mark.parametrize("flavour", ["musev", "musev_referencenet", "musev_referencenet_pose"])
def test_state_dict_keys_match_reference_inventory(flavour):
    """oracle.param_shapes is itself checked against the reference constructor by strict load_state_dict in
    tests/golden/make_reference_goldens.py"""
    from musev_amd.models.unet_loader import load_unet_by_name
    cfg = unet3d.flavour_config(flavour, **NARROW)
    want = unet3d.param_shapes(cfg)
    m = load_unet_by_name(flavour, dtype=torch.float32, **NARROW)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert set(got) == set(want)
    assert all(got[k] == tuple(want[k]) for k in want)
    # a seeded oracle state dict loads strictly (what real checkpoints would do)
    m.load_state_dict(unet3d.init_state_dict(cfg, 1), strict=True)


def test_full_size_parameter_count():
    shapes = unet3d.param_shapes(unet3d.flavour_config("musev"))
    n = sum(int(torch.tensor(s).prod()) for s in shapes.values())
    assert abs(n - 1.42e9) < 0.01e9   # SD-1.5 0.86 B + temporal convs + temporal transformers


def test_constructor_error_conventions():
    from musev_amd.models.unet_3d_condition import UNet3DConditionModel
    with pytest.raises(ValueError, match="same number of `down_block_types`"):
        UNet3DConditionModel(down_block_types=("DownBlock3D",), up_block_types=("UpBlock3D", "UpBlock3D"), block_out_channels=(64,))
    with pytest.raises(ValueError, match="same number of `block_out_channels`"):
        UNet3DConditionModel(block_out_channels=(64, 128))
    with pytest.raises(NotImplementedError, match="facein"):
        UNet3DConditionModel(block_out_channels=(64, 128, 256, 256), need_t2i_facein=True, ip_adapter_cross_attn=True)
    from musev_amd.models.unet_loader import load_unet_by_name
    with pytest.raises(ValueError, match="unsupport model_name"):
        load_unet_by_name("musev_v2")


def test_registry_and_pipeline_surface():
    from musev_amd.models import Model_Register
    from musev_amd.models.unet_loader import load_unet_by_name
    for name in ("TemporalConvLayer", "TransformerTemporalModel", "NonParamT2ISelfReferenceXFormersAttnProcessor",
                 "NonParamReferenceIPXFormersAttnProcessor", "T2IReferencenetIPAdapterXFormersAttnProcessor", "BaseIPAttnProcessor"):
        assert name in Model_Register
    m = load_unet_by_name("musev_referencenet", dtype=torch.float16, **NARROW)
    assert m.dtype == torch.float16 and m.device.type == "cpu"
    assert m.config.in_channels == 4 and m.ip_adapter_cross_attn is True
    attns, blocks = m.spatial_cross_attns
    assert len(attns) == 16 and all(hasattr(a, "to_k_ip") and hasattr(a, "to_v_ip") for _, a in attns)
    assert m.self_attn_num == 16
    m.set_skip_temporal_layers(True)
    assert all(getattr(x, "skip_temporal_layers") for x in m.modules() if hasattr(x, "skip_temporal_layers"))
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(2, 4, 2, 8, 8), 1, torch.zeros(2, 77, 768))


def test_bench_flop_counter_known_answers():
    """SURVEY.md 8d table"""
    import bench
    assert abs(bench.unet_flops(512, 512, 13, 2, "musev") / 1e12 - 36.95) < 0.01
    assert abs(bench.unet_flops(512, 512, 12, 2, "musev", n_vis=0) / 1e12 - 30.55) < 0.01
    assert abs(bench.unet_flops(512, 512, 13, 2, "musev_referencenet") / 1e12 - 41.34) < 0.01
    assert abs(bench.unet_flops(256, 256, 4, 2, "musev", n_vis=0) / 1e12 - 2.38) < 0.01


def test_referencenet_state_dict_and_surface():
    """musev_amd.models.referencenet.ReferenceNet2D owns exactly the parameters of the reference's ReferenceNet2D in block-
    embedding mode (oracle.referencenet.param_shapes is loaded strict=True into the reference constructor by
    tests/golden/make_reference_goldens.py), under the diffusers UNet2D key names; an SD checkpoint's decoder keys are
    ignored by the loader; the module refuses CPU tensors and the modes the shipped flavour does not use."""
    from oracle import referencenet as oref
    from musev_amd.models.referencenet import ReferenceNet2D, load_referencenet_by_name
    cfg = oref.referencenet_config(**NARROW)
    want = oref.param_shapes(cfg)
    sd = oref.init_state_dict(cfg, 2)
    full = dict(sd)
    full["up_blocks.0.resnets.0.conv1.weight"] = torch.zeros(1)  # decoder weights of a full SD UNet2D checkpoint
    full["conv_out.weight"] = torch.zeros(1)
    m = load_referencenet_by_name("musev_referencenet", full, dtype=torch.float32, **NARROW)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert set(got) == set(want)
    assert all(got[k] == tuple(want[k]) for k in want)
    assert all(torch.equal(m.state_dict()[k], sd[k]) for k in sd)
    assert m.need_block_embs and not m.need_self_attn_block_embs and m.dtype == torch.float32
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(2, 4, 8, 8), 0, torch.zeros(2, 4, 768), num_frames=1)
    with pytest.raises(NotImplementedError):
        ReferenceNet2D(need_self_attn_block_embs=True)
    with pytest.raises(ValueError, match="unsupport model_name"):
        load_referencenet_by_name("musev")
