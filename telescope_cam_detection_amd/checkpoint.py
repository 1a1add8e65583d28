"""Checkpoint key conversion: foreign RT-DETRv2 state dicts -> this build's tensor naming.

The reference loads `torch.load(path)['ema']['module']` (else `['model']`) into UPSTREAM's module tree
(/root/reference/src/rtdetr_detector.py:134-141); that tree (lyuwenyu/RT-DETR rtdetrv2_pytorch) is not in the
reference checkout, and HF transformers ships the same weights under its own key names.  This module maps both
layouts onto `weights.module_specs` names so that `fold_weights` / `pack_blob` can consume them:

* `hf_key_map(arch)` / `convert_hf_state(state, arch)` - HF `RTDetrV2ForObjectDetection` checkpoints
  (`model.safetensors` of e.g. PekingU/rtdetr_v2_r50vd).  VERIFIED offline: tests/test_checkpoint.py converts a randomly
  initialised HF model's state dict and requires the oracle to reproduce the HF forward (the same pin as oracle/make_golden.py).
* `convert_upstream_state(state, arch)` - upstream `.pth` key names, restated from upstream's module definitions
  (PResNet `conv1.conv1_k`, `res_layers.S.blocks.B.branch2{a,b,c}` / `short`, HybridEncoder `input_proj`, `encoder.0.layers.0`,
  `fpn_blocks`, RTDETRTransformerv2 `decoder.layers.I`, `dec_score_head`, `dec_bbox_head`, fused `in_proj_weight`).
  UNVERIFIED offline (no upstream checkpoint or source here): every tensor this build needs must be found with the expected
  shape, otherwise the conversion raises and names the first missing keys - it never guesses silently.

`load_foreign_state(path)` sniffs the layout.  Nothing here touches the GPU.
"""
from __future__ import annotations

from typing import Dict, Mapping, Optional, Tuple

from .arch import ARCHS, Arch
from .weights import backbone_blocks, block_has_shortcut, module_specs


def hf_key_map(arch: Arch) -> Dict[str, str]:
    """this build's un-fused tensor name -> HF state-dict key (HF:rt_detr_v2/modeling_rt_detr_v2.py module tree)."""
    m: Dict[str, str] = {}

    def conv(mine, hf_conv, hf_bn):
        m[mine + ".conv.w"] = hf_conv + ".weight"
        for s, h in zip("gbmv", ("weight", "bias", "running_mean", "running_var")):
            m[mine + ".bn." + s] = hf_bn + "." + h

    def lin(mine, hf):
        m[mine + ".w"] = hf + ".weight"
        m[mine + ".b"] = hf + ".bias"

    def ln(mine, hf):
        m[mine + ".g"] = hf + ".weight"
        m[mine + ".b"] = hf + ".bias"

    bb = "model.backbone.model."
    for i in range(3):
        conv(f"backbone.stem.{i}", f"{bb}embedder.embedder.{i}.convolution", f"{bb}embedder.embedder.{i}.normalization")
    for pfx, cin, cout, stride, first in backbone_blocks(arch):
        _, s, b = pfx.split(".")
        base = f"{bb}encoder.stages.{s[1:]}.layers.{b[1:]}"
        n = 3 if arch.layer_type == "bottleneck" else 2
        for j in range(n):
            conv(f"{pfx}.c{j + 1}", f"{base}.layer.{j}.convolution", f"{base}.layer.{j}.normalization")
        if block_has_shortcut(arch, cin, cout, stride, first):
            sc = f"{base}.shortcut.1" if stride == 2 else f"{base}.shortcut"
            conv(pfx + ".sc", sc + ".convolution", sc + ".normalization")
    for l in range(3):
        conv(f"enc.proj.{l}", f"model.encoder_input_proj.{l}.0", f"model.encoder_input_proj.{l}.1")
        conv(f"dec.proj.{l}", f"model.decoder_input_proj.{l}.0", f"model.decoder_input_proj.{l}.1")
    a = "model.encoder.aifi.0.layers.0."
    for n in "qkvo":
        lin(f"enc.aifi.{n}", f"{a}self_attn.{n}_proj")
    ln("enc.aifi.ln1", a + "self_attn_layer_norm")
    lin("enc.aifi.fc1", a + "mlp.fc1")
    lin("enc.aifi.fc2", a + "mlp.fc2")
    ln("enc.aifi.ln2", a + "final_layer_norm")

    def csp(mine, hf):
        for c in ("1", "2"):
            conv(f"{mine}.c{c}", f"{hf}.conv{c}.conv", f"{hf}.conv{c}.norm")
        for j in range(3):
            conv(f"{mine}.rep{j}.k3", f"{hf}.bottlenecks.{j}.conv1.conv", f"{hf}.bottlenecks.{j}.conv1.norm")
            conv(f"{mine}.rep{j}.k1", f"{hf}.bottlenecks.{j}.conv2.conv", f"{hf}.bottlenecks.{j}.conv2.norm")
        if arch.csp_hidden != arch.enc_dim:
            conv(f"{mine}.c3", f"{hf}.conv3.conv", f"{hf}.conv3.norm")

    for i in range(2):
        conv(f"enc.lat.{i}", f"model.encoder.lateral_convs.{i}.conv", f"model.encoder.lateral_convs.{i}.norm")
        conv(f"enc.down.{i}", f"model.encoder.downsample_convs.{i}.conv", f"model.encoder.downsample_convs.{i}.norm")
        csp(f"enc.fpn.{i}", f"model.encoder.fpn_blocks.{i}")
        csp(f"enc.pan.{i}", f"model.encoder.pan_blocks.{i}")
    lin("dec.enc_out.fc", "model.enc_output.0")
    ln("dec.enc_out.ln", "model.enc_output.1")
    lin("dec.enc_score", "model.enc_score_head")
    for j in range(3):
        lin(f"dec.enc_bbox.{j}", f"model.enc_bbox_head.layers.{j}")
    for j in range(2):
        lin(f"dec.qpos.{j}", f"model.decoder.query_pos_head.layers.{j}")
    for i in range(arch.dec_layers):
        d = f"model.decoder.layers.{i}."
        for n in "qkvo":
            lin(f"dec.l{i}.sa.{n}", f"{d}self_attn.{n}_proj")
        ln(f"dec.l{i}.ln1", d + "self_attn_layer_norm")
        lin(f"dec.l{i}.ca.off", d + "encoder_attn.sampling_offsets")
        lin(f"dec.l{i}.ca.aw", d + "encoder_attn.attention_weights")
        lin(f"dec.l{i}.ca.vp", d + "encoder_attn.value_proj")
        lin(f"dec.l{i}.ca.op", d + "encoder_attn.output_proj")
        ln(f"dec.l{i}.ln2", d + "encoder_attn_layer_norm")
        lin(f"dec.l{i}.fc1", d + "mlp.fc1")
        lin(f"dec.l{i}.fc2", d + "mlp.fc2")
        ln(f"dec.l{i}.ln3", d + "final_layer_norm")
        for j in range(3):
            lin(f"dec.bbox.{i}.{j}", f"bbox_embed.{i}.layers.{j}")
        lin(f"dec.cls.{i}", f"class_embed.{i}")
    return m


def expected_shapes(arch: Arch) -> Dict[str, Tuple[int, ...]]:
    """shape of every un-fused tensor this build loads (from weights.module_specs)"""
    convs, lins, lns = module_specs(arch)
    sh: Dict[str, Tuple[int, ...]] = {}
    for c in convs:
        sh[c.name + ".conv.w"] = (c.cout, c.cin, c.k, c.k)
        for s in "gbmv":
            sh[c.name + ".bn." + s] = (c.cout,)
    for l in lins:
        sh[l.name + ".w"] = (l.cout, l.cin)
        sh[l.name + ".b"] = (l.cout,)
    for n in lns:
        sh[n.name + ".g"] = (n.dim,)
        sh[n.name + ".b"] = (n.dim,)
    return sh


def _finish(out: Dict[str, "object"], arch: Arch, what: str):
    sh = expected_shapes(arch)
    missing = [k for k in sh if k not in out]
    if missing:
        raise KeyError(f"{what}: {len(missing)} tensors of arch '{arch.name}' not found, first: {missing[:6]}")
    bad = [(k, tuple(out[k].shape), sh[k]) for k in sh if tuple(out[k].shape) != sh[k]]
    if bad:
        raise ValueError(f"{what}: shape mismatch (name, got, want), first: {bad[:4]}")
    return {k: out[k].detach().float().contiguous() for k in sh}


def convert_hf_state(state: Mapping[str, "object"], arch: Arch):
    """HF `RTDetrV2ForObjectDetection.state_dict()` -> this build's naming (all tensors, shape-checked)."""
    km = hf_key_map(arch)
    out = {}
    for mine, hf in km.items():
        if hf in state:
            out[mine] = state[hf]
        elif "model.decoder." + hf in state:          # heads saved only under their decoder alias
            out[mine] = state["model.decoder." + hf]
    return _finish(out, arch, "HF checkpoint")


def upstream_key_map(arch: Arch) -> Dict[str, str]:
    """this build's tensor name -> UPSTREAM (lyuwenyu rtdetrv2_pytorch) key.  Entries ending in '#q', '#k', '#v' are thirds of a
    fused `in_proj_weight` / `in_proj_bias`.  Unverified offline - see the module docstring."""
    m: Dict[str, str] = {}

    def conv(mine, up):                                # upstream ConvNormLayer: .conv / .norm
        m[mine + ".conv.w"] = up + ".conv.weight"
        for s, h in zip("gbmv", ("weight", "bias", "running_mean", "running_var")):
            m[mine + ".bn." + s] = up + ".norm." + h

    def lin(mine, up):
        m[mine + ".w"] = up + ".weight"
        m[mine + ".b"] = up + ".bias"

    def ln(mine, up):
        m[mine + ".g"] = up + ".weight"
        m[mine + ".b"] = up + ".bias"

    def mha(mine, up):                                 # nn.MultiheadAttention
        for n in "qkv":
            m[f"{mine}.{n}.w"] = f"{up}.in_proj_weight#{n}"
            m[f"{mine}.{n}.b"] = f"{up}.in_proj_bias#{n}"
        lin(mine + ".o", up + ".out_proj")

    for i in range(3):
        conv(f"backbone.stem.{i}", f"backbone.conv1.conv1_{i + 1}")
    branches = ("branch2a", "branch2b", "branch2c")
    for pfx, cin, cout, stride, first in backbone_blocks(arch):
        _, s, b = pfx.split(".")
        base = f"backbone.res_layers.{s[1:]}.blocks.{b[1:]}"
        n = 3 if arch.layer_type == "bottleneck" else 2
        for j in range(n):
            conv(f"{pfx}.c{j + 1}", f"{base}.{branches[j]}")
        if block_has_shortcut(arch, cin, cout, stride, first):
            conv(pfx + ".sc", f"{base}.short.conv" if stride == 2 else f"{base}.short")   # variant d: AvgPool + ConvNorm named 'conv'
    for l in range(3):
        m[f"enc.proj.{l}.conv.w"] = f"encoder.input_proj.{l}.conv.weight"
        m[f"dec.proj.{l}.conv.w"] = f"decoder.input_proj.{l}.conv.weight"
        for s, h in zip("gbmv", ("weight", "bias", "running_mean", "running_var")):
            m[f"enc.proj.{l}.bn.{s}"] = f"encoder.input_proj.{l}.norm.{h}"
            m[f"dec.proj.{l}.bn.{s}"] = f"decoder.input_proj.{l}.norm.{h}"
    a = "encoder.encoder.0.layers.0"
    mha("enc.aifi", a + ".self_attn")
    ln("enc.aifi.ln1", a + ".norm1")
    lin("enc.aifi.fc1", a + ".linear1")
    lin("enc.aifi.fc2", a + ".linear2")
    ln("enc.aifi.ln2", a + ".norm2")

    def csp(mine, up):
        conv(f"{mine}.c1", f"{up}.conv1")
        conv(f"{mine}.c2", f"{up}.conv2")
        for j in range(3):
            conv(f"{mine}.rep{j}.k3", f"{up}.bottlenecks.{j}.conv1")
            conv(f"{mine}.rep{j}.k1", f"{up}.bottlenecks.{j}.conv2")
        if arch.csp_hidden != arch.enc_dim:
            conv(f"{mine}.c3", f"{up}.conv3")

    for i in range(2):
        conv(f"enc.lat.{i}", f"encoder.lateral_convs.{i}")
        conv(f"enc.down.{i}", f"encoder.downsample_convs.{i}")
        csp(f"enc.fpn.{i}", f"encoder.fpn_blocks.{i}")
        csp(f"enc.pan.{i}", f"encoder.pan_blocks.{i}")
    lin("dec.enc_out.fc", "decoder.enc_output.proj")
    ln("dec.enc_out.ln", "decoder.enc_output.norm")
    lin("dec.enc_score", "decoder.enc_score_head")
    for j in range(3):
        lin(f"dec.enc_bbox.{j}", f"decoder.enc_bbox_head.layers.{j}")
    for j in range(2):
        lin(f"dec.qpos.{j}", f"decoder.query_pos_head.layers.{j}")
    for i in range(arch.dec_layers):
        d = f"decoder.decoder.layers.{i}"
        mha(f"dec.l{i}.sa", d + ".self_attn")
        ln(f"dec.l{i}.ln1", d + ".norm1")
        lin(f"dec.l{i}.ca.off", d + ".cross_attn.sampling_offsets")
        lin(f"dec.l{i}.ca.aw", d + ".cross_attn.attention_weights")
        lin(f"dec.l{i}.ca.vp", d + ".cross_attn.value_proj")
        lin(f"dec.l{i}.ca.op", d + ".cross_attn.output_proj")
        ln(f"dec.l{i}.ln2", d + ".norm2")
        lin(f"dec.l{i}.fc1", d + ".linear1")
        lin(f"dec.l{i}.fc2", d + ".linear2")
        ln(f"dec.l{i}.ln3", d + ".norm3")
        for j in range(3):
            lin(f"dec.bbox.{i}.{j}", f"decoder.dec_bbox_head.{i}.layers.{j}")
        lin(f"dec.cls.{i}", f"decoder.dec_score_head.{i}")
    return m


def convert_upstream_state(state: Mapping[str, "object"], arch: Arch):
    """upstream `ckpt['ema']['module']` / `ckpt['model']` -> this build's naming.  Raises when a tensor is absent or has the wrong
    shape (the key table is unverified offline)."""
    km = upstream_key_map(arch)
    state = {(k[7:] if k.startswith("module.") else k): v for k, v in state.items()}
    out = {}
    for mine, up in km.items():
        if "#" in up:
            key, part = up.split("#")
            if key in state:
                t = state[key]
                third = t.shape[0] // 3
                out[mine] = t["qkv".index(part) * third:("qkv".index(part) + 1) * third]
        elif up in state:
            out[mine] = state[up]
    return _finish(out, arch, "upstream checkpoint")


def sniff_layout(state: Mapping[str, "object"]) -> str:
    keys = list(state.keys())
    if any(k.startswith("model.backbone.model.") for k in keys):
        return "hf"
    if any(k.startswith("backbone.conv1.") or k.startswith("module.backbone.conv1.") for k in keys):
        return "upstream"
    if any(k.startswith("backbone.stem.") for k in keys):
        return "native"
    raise ValueError("unrecognised RT-DETR checkpoint layout (first keys: %s)" % keys[:4])


def guess_arch(state: Mapping[str, "object"], layout: str) -> Optional[str]:
    """pick the ARCHS entry whose shapes fit (r18 / r34 / r50 / r101 differ in block counts and widths)"""
    conv = {"hf": convert_hf_state, "upstream": convert_upstream_state}.get(layout)
    for name, arch in ARCHS.items():
        try:
            if conv is not None:
                conv(state, arch)
            else:
                _finish(dict(state), arch, "native")
            return name
        except (KeyError, ValueError):
            continue
    return None


def load_foreign_state(path: str, arch_name: Optional[str] = None):
    """Read `.safetensors` (HF) or a torch checkpoint (HF `pytorch_model.bin`, upstream `.pth`, this build's own format) and return
    (state in this build's naming, arch name)."""
    if str(path).endswith(".safetensors"):
        from safetensors.torch import load_file

        state = load_file(path)
        hint = None
    else:
        import torch

        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        hint = ckpt.get("arch") if isinstance(ckpt, dict) else None
        if isinstance(ckpt, dict) and "ema" in ckpt and isinstance(ckpt["ema"], dict) and "module" in ckpt["ema"]:
            state = ckpt["ema"]["module"]              # src/rtdetr_detector.py:135-136
        elif isinstance(ckpt, dict) and "model" in ckpt and isinstance(ckpt["model"], dict):
            state = ckpt["model"]                      # :138
        else:
            state = ckpt
    layout = sniff_layout(state)
    name = arch_name or hint or guess_arch(state, layout)
    if name is None:
        raise ValueError("could not match the checkpoint to r18 / r34 / r50 / r101; pass the arch explicitly")
    arch = ARCHS[name]
    if layout == "hf":
        return convert_hf_state(state, arch), name
    if layout == "upstream":
        return convert_upstream_state(state, arch), name
    return _finish(dict(state), arch, "native checkpoint"), name
