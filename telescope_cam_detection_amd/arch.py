"""RT-DETRv2 architecture descriptors (R18 / R34 / R50 / R101).

The reference never spells the network out: `src/rtdetr_detector.py:132` hands a YAML
path to the un-vendored lyuwenyu/RT-DETR `YAMLConfig`.  The hyper-parameters below are
the ones SURVEY.md §8(c) cross-checked against upstream's published parameter / FLOP
counts (20 M / 42 M / 76 M params; 60 / 136 GFLOPs @640²).

An `Arch` is pure data.  It is serialised into `rtd_config` (include/rtdetr_mi355.h) for
the HIP engine and drives the weight recipe / packer (weights.py) and the CPU oracle
(oracle/rtdetr_oracle.py), so all three agree on one description of the graph.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple


@dataclass(frozen=True)
class Arch:
    name: str
    # PResNet-vd backbone (HF:rt_detr/modeling_rt_detr_resnet.py:71-310)
    layer_type: str                      # "basic" | "bottleneck"
    depths: Tuple[int, int, int, int]
    hidden_sizes: Tuple[int, int, int, int]
    embedding_size: int = 64
    # hybrid encoder (HF:rt_detr_v2/modeling_rt_detr_v2.py:1098-1209)
    enc_dim: int = 256
    enc_ffn: int = 1024
    enc_heads: int = 8
    expansion: float = 1.0               # CSPRep hidden = int(enc_dim * expansion)
    # decoder (HF:...v2.py:555-661, 1339-1623)
    d_model: int = 256
    dec_ffn: int = 1024
    dec_heads: int = 8
    dec_layers: int = 6
    num_queries: int = 300
    num_classes: int = 80
    n_levels: int = 3
    n_points: int = 4
    offset_scale: float = 0.5
    feat_strides: Tuple[int, int, int] = (8, 16, 32)

    @property
    def backbone_out_channels(self) -> Tuple[int, int, int]:
        return tuple(self.hidden_sizes[1:])  # stages 2..4 (out_indices [2,3,4])

    @property
    def csp_hidden(self) -> int:
        return int(self.enc_dim * self.expansion)

    def level_shapes(self, height: int, width: int) -> List[Tuple[int, int]]:
        """(h, w) of the stride-8/16/32 maps for an input of height x width.

        Follows the conv arithmetic of the stem (3x3 s2 p1, maxpool 3x3 s2 p1) and the
        stride-2 stages (3x3 s2 p1 on the main path, AvgPool2d(2,2,ceil) on the shortcut).
        """
        def down(n):  # k=3, s=2, p=1
            return (n + 2 - 3) // 2 + 1
        h, w = down(down(height)), down(down(width))  # stride 4
        out = []
        for _ in range(3):
            h, w = down(h), down(w)
            out.append((h, w))
        return out


ARCHS = {
    "r18": Arch("r18", "basic", (2, 2, 2, 2), (64, 128, 256, 512), expansion=0.5, dec_layers=3),
    "r34": Arch("r34", "basic", (3, 4, 6, 3), (64, 128, 256, 512), expansion=0.5, dec_layers=4),
    "r50": Arch("r50", "bottleneck", (3, 4, 6, 3), (256, 512, 1024, 2048)),
    "r101": Arch("r101", "bottleneck", (3, 4, 23, 3), (256, 512, 1024, 2048), enc_dim=384, enc_ffn=2048),
    # tiny graph-shaped model for fast CPU/GPU unit tests (not a reference variant)
    "tiny": Arch("tiny", "bottleneck", (1, 1, 1, 1), (64, 128, 256, 512), embedding_size=32,
                 enc_dim=64, enc_ffn=128, enc_heads=2, d_model=64, dec_ffn=128, dec_heads=2,
                 dec_layers=2, num_queries=50, expansion=1.0),
    "tinyb": Arch("tinyb", "basic", (1, 1, 1, 1), (32, 64, 128, 256), embedding_size=32,
                  enc_dim=64, enc_ffn=128, enc_heads=2, d_model=64, dec_ffn=128, dec_heads=2,
                  dec_layers=2, num_queries=50, expansion=0.5),
    # a one-block-per-stage R18 trunk with small heads: every trunk width is a whole 32-channel group, which the f16x3 engine's hi/lo
    # layout needs (basic blocks need embedding_size == hidden_sizes[0], HF:rt_detr_resnet.py:150-163)
    "tinyc": Arch("tinyc", "basic", (1, 1, 1, 1), (64, 128, 256, 512), embedding_size=64,
                  enc_dim=64, enc_ffn=128, enc_heads=2, d_model=64, dec_ffn=128, dec_heads=2,
                  dec_layers=2, num_queries=50, expansion=0.5),
}


def arch_from_config_path(config_path: str) -> Arch:
    """Map the reference's `config_path` argument (src/rtdetr_detector.py:31) to a variant.

    The reference passes e.g. "RT-DETR/rtdetrv2_pytorch/configs/rtdetrv2/rtdetrv2_r18vd_120e_coco.yml";
    upstream's file names carry the backbone tag, which is all this build needs from the YAML.
    """
    s = str(config_path).lower()
    for tag in ("r101", "r50", "r34", "r18", "tinyc", "tinyb", "tiny"):
        if tag in s:
            return ARCHS[tag]
    raise ValueError(f"cannot infer RT-DETR variant from config_path={config_path!r}")
