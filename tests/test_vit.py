"""DINOv2 forward: the float32 oracle is pinned against the `transformers` Dinov2 architecture
shipped in this image (random weights from a config object), and the product module is checked
against the oracle on CPU (float32).  GPU bf16 accuracy is in tests/test_vit_gpu.py."""
import numpy as np
import pytest
import torch

from oracle import vit_oracle


def hf_state_dict_to_dinov2(hf_sd, depth):
    sd = {
        "cls_token": hf_sd["embeddings.cls_token"],
        "pos_embed": hf_sd["embeddings.position_embeddings"],
        "mask_token": hf_sd["embeddings.mask_token"],
        "patch_embed.proj.weight": hf_sd["embeddings.patch_embeddings.projection.weight"],
        "patch_embed.proj.bias": hf_sd["embeddings.patch_embeddings.projection.bias"],
        "norm.weight": hf_sd["layernorm.weight"],
        "norm.bias": hf_sd["layernorm.bias"],
    }
    for i in range(depth):
        h, p = f"encoder.layer.{i}.", f"blocks.{i}."
        for n in ("norm1", "norm2"):
            sd[p + n + ".weight"] = hf_sd[h + n + ".weight"]
            sd[p + n + ".bias"] = hf_sd[h + n + ".bias"]
        a = h + "attention.attention."
        sd[p + "attn.qkv.weight"] = torch.cat([hf_sd[a + "query.weight"], hf_sd[a + "key.weight"], hf_sd[a + "value.weight"]])
        sd[p + "attn.qkv.bias"] = torch.cat([hf_sd[a + "query.bias"], hf_sd[a + "key.bias"], hf_sd[a + "value.bias"]])
        sd[p + "attn.proj.weight"] = hf_sd[h + "attention.output.dense.weight"]
        sd[p + "attn.proj.bias"] = hf_sd[h + "attention.output.dense.bias"]
        sd[p + "ls1.gamma"] = hf_sd[h + "layer_scale1.lambda1"]
        sd[p + "ls2.gamma"] = hf_sd[h + "layer_scale2.lambda1"]
        for n in ("fc1", "fc2"):
            sd[p + f"mlp.{n}.weight"] = hf_sd[h + f"mlp.{n}.weight"]
            sd[p + f"mlp.{n}.bias"] = hf_sd[h + f"mlp.{n}.bias"]
    return {k: v.detach().clone().float() for k, v in sd.items()}


def make_hf(hidden, depth, heads, seed):
    from transformers import Dinov2Config, Dinov2Model

    torch.manual_seed(seed)
    cfg = Dinov2Config(hidden_size=hidden, num_hidden_layers=depth, num_attention_heads=heads, mlp_ratio=4,
                       patch_size=14, image_size=518, layerscale_value=1.0)
    m = Dinov2Model(cfg).eval()
    with torch.no_grad():  # make every parameter non-trivial (layer scale, biases, norms)
        g = torch.Generator().manual_seed(seed + 1)
        for n, p in m.named_parameters():
            if "lambda1" in n:
                p.copy_(0.5 + torch.rand(p.shape, generator=g))
            elif p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    return m


@pytest.mark.parametrize("hidden,depth,heads,hw", [(64, 2, 4, (70, 98)), (384, 2, 6, (476, 630))])
def test_oracle_equals_transformers_dinov2(hidden, depth, heads, hw):
    hf = make_hf(hidden, depth, heads, seed=3)
    sd = hf_state_dict_to_dinov2(hf.state_dict(), depth)
    torch.manual_seed(0)
    x = torch.randn(2, 3, *hw)
    with torch.no_grad():
        ref = hf(pixel_values=x).last_hidden_state[:, 1:]
        got = vit_oracle.forward_patch_tokens(sd, x, heads, interpolate_offset=0.0)   # transformers: size-based resize
    assert got.shape == ref.shape == (2, (hw[0] // 14) * (hw[1] // 14), hidden)
    torch.testing.assert_close(got, ref, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("name,offset", [("dinov2_vits14", 0.1), ("dinov2_vits14", 0.0), ("dinov2_vits14_reg", 0.1)])
def test_product_module_equals_oracle_on_cpu(name, offset):
    from vit_colmap_amd.vit import build_dinov2

    torch.manual_seed(0)
    model = build_dinov2(name, interpolate_offset=offset).init_random(seed=5).eval()
    with torch.no_grad():
        for b in model.blocks[:3]:
            b.ls1.gamma.mul_(0.7)
            b.ls2.gamma.mul_(1.3)
    model.blocks = model.blocks[:3]                      # 3 layers keep the CPU test quick
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.randn(1, 3, 224, 308)
    with torch.no_grad():
        ref = vit_oracle.forward_patch_tokens(sd, x, model.arch.heads, interpolate_offset=offset)
        got = model.forward_features(x)["x_norm_patchtokens"]
        torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)
        model.fold_layerscale()
        folded = model.forward_features(x)["x_norm_patchtokens"]
        torch.testing.assert_close(folded, ref, rtol=1e-4, atol=1e-4)


def test_architectures_and_errors():
    from vit_colmap_amd.vit import DINOV2_ARCHS, build_dinov2

    assert DINOV2_ARCHS["dinov2_vits14"].dim == 384 and DINOV2_ARCHS["dinov2_vitb14"].dim == 768
    m = build_dinov2("dinov2_vits14")
    n_params = sum(p.numel() for n, p in m.named_parameters() if n != "mask_token")
    assert abs(n_params - 22.06e6) < 0.05e6                 # 22.06 M parameters
    with pytest.raises(ValueError):
        build_dinov2("resnet50")                            # vit_extractor.py:100-104


def test_checkpoint_roundtrip(tmp_path):
    from vit_colmap_amd.vit import build_dinov2, load_dinov2_weights

    a = build_dinov2("dinov2_vits14").init_random(seed=1)
    path = tmp_path / "w.pth"
    torch.save(a.state_dict(), path)
    b = load_dinov2_weights(build_dinov2("dinov2_vits14"), str(path))
    for (n1, p1), (n2, p2) in zip(a.state_dict().items(), b.state_dict().items()):
        assert n1 == n2 and torch.equal(p1, p2)
    with pytest.raises(ValueError):
        load_dinov2_weights(build_dinov2("dinov2_vitb14"), str(path))
