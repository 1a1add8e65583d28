"""CPU: checkpoint key conversion (telescope_cam_detection_amd/checkpoint.py)."""
import os

import numpy as np
import pytest
import torch

from telescope_cam_detection_amd import checkpoint as ck
from telescope_cam_detection_amd.arch import ARCHS
from telescope_cam_detection_amd.weights import synth_weights


def _hf_model(arch):
    tf = pytest.importorskip("transformers")
    from transformers import RTDetrV2Config, RTDetrV2ForObjectDetection
    from transformers.models.rt_detr.configuration_rt_detr_resnet import RTDetrResNetConfig

    bb = RTDetrResNetConfig(depths=list(arch.depths), layer_type=arch.layer_type, hidden_sizes=list(arch.hidden_sizes),
                            embedding_size=arch.embedding_size, out_indices=[2, 3, 4])
    cfg = RTDetrV2Config(
        backbone_config=bb, encoder_in_channels=list(arch.backbone_out_channels), encoder_hidden_dim=arch.enc_dim,
        encoder_ffn_dim=arch.enc_ffn, encoder_attention_heads=arch.enc_heads, hidden_expansion=arch.expansion,
        d_model=arch.d_model, decoder_in_channels=[arch.enc_dim] * 3, decoder_ffn_dim=arch.dec_ffn,
        decoder_attention_heads=arch.dec_heads, decoder_layers=arch.dec_layers, num_queries=arch.num_queries,
        num_labels=arch.num_classes, anchor_image_size=None, eval_size=None, tie_word_embeddings=False)
    cfg._attn_implementation = "eager"
    torch.manual_seed(7)
    model = RTDetrV2ForObjectDetection(cfg).eval()
    # HF's default init leaves BatchNorm at identity and the score heads at a constant bias: make every tensor distinctive
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for name, p in list(model.named_parameters()) + list(model.named_buffers()):
            if "num_batches_tracked" in name or "n_points_scale" in name or "anchors" in name or "valid_mask" in name:
                continue
            if "running_var" in name:
                p.copy_(0.5 + torch.rand(p.shape, generator=g))
            elif p.dim() <= 1:
                p.add_(0.05 * torch.randn(p.shape, generator=g))
    return model


def test_hf_state_dict_converts_and_reproduces_the_hf_forward():
    """A (randomly initialised) HF RTDetrV2ForObjectDetection checkpoint, converted key by key, must make the oracle reproduce
    HF's own logits and boxes - i.e. every tensor landed in the right slot."""
    from oracle import rtdetr_oracle as orc

    arch = ARCHS["tiny"]
    model = _hf_model(arch)
    w = ck.convert_hf_state(model.state_dict(), arch)
    assert set(w) == set(synth_weights(arch, 0))
    x = torch.rand(1, 3, 160, 160, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        ref = model(pixel_values=x)
    col = {}
    orc.model_forward(arch, w, x, [(160, 160)], collect=col)
    # the 300 selected queries may be ordered differently on near-ties: compare per selected memory token
    hf_topk = None
    got_logits, got_boxes = col["logits"][0].numpy(), col["pred_boxes"][0].numpy()
    want_logits, want_boxes = ref.logits[0].numpy(), ref.pred_boxes[0].numpy()
    # match rows by nearest box+logit signature
    sig_g = np.concatenate([got_boxes, got_logits[:, :4]], 1)
    sig_w = np.concatenate([want_boxes, want_logits[:, :4]], 1)
    d = np.abs(sig_g[:, None, :] - sig_w[None, :, :]).max(-1)
    assert (d.min(1) < 2e-4).mean() > 0.99, float(d.min(1).max())


def test_safetensors_file_is_sniffed_and_loaded(tmp_path):
    st = pytest.importorskip("safetensors.torch")
    arch = ARCHS["tiny"]
    w = synth_weights(arch, 5)
    km = ck.hf_key_map(arch)
    path = os.path.join(tmp_path, "model.safetensors")
    st.save_file({hf: w[mine].contiguous() for mine, hf in km.items()}, path)
    got, name = ck.load_foreign_state(path, arch_name="tiny")
    assert name == "tiny" and all(torch.equal(got[k], w[k]) for k in w)
    # arch inferred from the shapes when not given
    got2, name2 = ck.load_foreign_state(path)
    assert name2 in ("tiny",) and all(torch.equal(got2[k], w[k]) for k in w)


def test_upstream_names_round_trip_with_fused_in_proj(tmp_path):
    """Upstream layout (unverified key table): a state dict written with the table (q/k/v fused into in_proj_*) converts back
    exactly; a missing tensor is reported, not guessed."""
    arch = ARCHS["r18"]
    w = synth_weights(arch, 2)
    km = ck.upstream_key_map(arch)
    up, fused = {}, {}
    for mine, key in km.items():
        if "#" in key:
            base, part = key.split("#")
            fused.setdefault(base, {})[part] = w[mine]
        else:
            up["module." + key] = w[mine]
    for base, parts in fused.items():
        up["module." + base] = torch.cat([parts["q"], parts["k"], parts["v"]], 0)
    path = os.path.join(tmp_path, "rtdetrv2_r18vd.pth")
    torch.save({"ema": {"module": up}}, path)
    got, name = ck.load_foreign_state(path)
    assert name == "r18" and all(torch.equal(got[k], w[k]) for k in w)
    del up["module.decoder.dec_score_head.2.weight"]
    with pytest.raises(KeyError, match="dec.cls.2.w"):
        ck.convert_upstream_state(up, arch)


def test_native_file_still_loads(tmp_path):
    from telescope_cam_detection_amd.rtdetr_detector import load_state
    from telescope_cam_detection_amd.weights import save_weights

    arch = ARCHS["tiny"]
    w = synth_weights(arch, 9)
    path = os.path.join(tmp_path, "native.pth")
    save_weights(path, arch, w)
    got, name = load_state(path)
    assert name == "tiny" and all(torch.equal(got[k], w[k]) for k in w)
