"""Weight description, synthetic weight recipe, load-time folding and the flat weight blob.

Three things live here, all host-side "load_model" work (reference: src/rtdetr_detector.py:132-173):

* `module_specs(arch)`  - the un-fused parameter inventory of RT-DETRv2 in this build's own
  naming (conv+BN pairs, linears, layer norms).  The oracle consumes exactly these tensors.
* `synth_weights(arch, seed)` - the deterministic re-conditioned random weights SURVEY.md §8(c)
  prescribes (no checkpoint exists offline; HF's default init gives an all-ties top-k).
* `fold_weights()` / `pack_blob()` - what upstream's `.deploy()` does at load time
  (src/rtdetr_detector.py:164-165): BN folded into the conv, RepVGG 3x3+1x1 re-parameterised
  into one 3x3, plus this build's own layout choices (OHWI filters, fused sibling GEMMs).  The blob is a flat
  container of named fp32 tensors that `rtd_load_weights` (include/rtdetr_mi355.h) parses.
"""
from __future__ import annotations

import struct
import zlib
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np
import torch

from .arch import Arch

BN_EPS = 1e-5  # HF:rt_detr_v2/modeling_rt_detr_v2.py:748 (frozen BN), config.batch_norm_eps


@dataclass(frozen=True)
class ConvSpec:
    name: str
    cin: int
    cout: int
    k: int
    gain: float = 1.0


@dataclass(frozen=True)
class LinSpec:
    name: str
    cin: int
    cout: int
    kind: str = "lin"   # lin | score | small | offset


@dataclass(frozen=True)
class LNSpec:
    name: str
    dim: int


def backbone_blocks(arch: Arch):
    """Yield (prefix, cin, cout, stride, first) for every residual block.

    HF:rt_detr/modeling_rt_detr_resnet.py:229-310 (stage 0 has stride 1, stages 1-3 stride 2).
    """
    cin = arch.embedding_size
    for si, (cout, depth) in enumerate(zip(arch.hidden_sizes, arch.depths)):
        for bi in range(depth):
            stride = 2 if (si > 0 and bi == 0) else 1
            yield f"backbone.s{si}.b{bi}", cin, cout, stride, bi == 0
            cin = cout


def block_has_shortcut(arch: Arch, cin: int, cout: int, stride: int, first: bool) -> bool:
    if arch.layer_type == "bottleneck":
        return cin != cout or stride != 1          # resnet.py:197
    return first                                    # resnet.py:254 (should_apply_shortcut=True)


def module_specs(arch: Arch):
    convs: List[ConvSpec] = []
    lins: List[LinSpec] = []
    lns: List[LNSpec] = []
    e = arch.embedding_size
    # stem, resnet.py:71-114
    convs += [ConvSpec("backbone.stem.0", 3, e // 2, 3), ConvSpec("backbone.stem.1", e // 2, e // 2, 3),
              ConvSpec("backbone.stem.2", e // 2, e, 3)]
    for pfx, cin, cout, stride, first in backbone_blocks(arch):
        if arch.layer_type == "bottleneck":
            mid = cout // 4
            convs += [ConvSpec(pfx + ".c1", cin, mid, 1), ConvSpec(pfx + ".c2", mid, mid, 3),
                      ConvSpec(pfx + ".c3", mid, cout, 1, gain=0.5)]
        else:
            convs += [ConvSpec(pfx + ".c1", cin, cout, 3), ConvSpec(pfx + ".c2", cout, cout, 3, gain=0.5)]
        if block_has_shortcut(arch, cin, cout, stride, first):
            convs.append(ConvSpec(pfx + ".sc", cin, cout, 1, gain=0.7))
    d = arch.enc_dim
    for l, c in enumerate(arch.backbone_out_channels):
        convs.append(ConvSpec(f"enc.proj.{l}", c, d, 1))
    for n in "qkvo":
        lins.append(LinSpec(f"enc.aifi.{n}", d, d))
    lns.append(LNSpec("enc.aifi.ln1", d))
    lins += [LinSpec("enc.aifi.fc1", d, arch.enc_ffn), LinSpec("enc.aifi.fc2", arch.enc_ffn, d)]
    lns.append(LNSpec("enc.aifi.ln2", d))
    h = arch.csp_hidden

    def csp(pfx):
        out = [ConvSpec(pfx + ".c1", 2 * d, h, 1), ConvSpec(pfx + ".c2", 2 * d, h, 1)]
        for j in range(3):
            out += [ConvSpec(f"{pfx}.rep{j}.k3", h, h, 3, gain=0.7), ConvSpec(f"{pfx}.rep{j}.k1", h, h, 1, gain=0.7)]
        if h != d:
            out.append(ConvSpec(pfx + ".c3", h, d, 1))
        return out

    for i in range(2):
        convs.append(ConvSpec(f"enc.lat.{i}", d, d, 1))
        convs += csp(f"enc.fpn.{i}")
    for i in range(2):
        convs.append(ConvSpec(f"enc.down.{i}", d, d, 3))
        convs += csp(f"enc.pan.{i}")
    dm = arch.d_model
    for l in range(arch.n_levels):
        convs.append(ConvSpec(f"dec.proj.{l}", d, dm, 1))
    lins.append(LinSpec("dec.enc_out.fc", dm, dm))
    lns.append(LNSpec("dec.enc_out.ln", dm))
    lins.append(LinSpec("dec.enc_score", dm, arch.num_classes, "score"))
    lins += [LinSpec("dec.enc_bbox.0", dm, dm), LinSpec("dec.enc_bbox.1", dm, dm),
             LinSpec("dec.enc_bbox.2", dm, 4, "small")]
    lins += [LinSpec("dec.qpos.0", 4, 2 * dm), LinSpec("dec.qpos.1", 2 * dm, dm)]
    npts = arch.dec_heads * arch.n_levels * arch.n_points
    for i in range(arch.dec_layers):
        p = f"dec.l{i}"
        for n in "qkvo":
            lins.append(LinSpec(f"{p}.sa.{n}", dm, dm))
        lns.append(LNSpec(f"{p}.ln1", dm))
        lins += [LinSpec(f"{p}.ca.off", dm, npts * 2, "offset"), LinSpec(f"{p}.ca.aw", dm, npts, "small"),
                 LinSpec(f"{p}.ca.vp", dm, dm), LinSpec(f"{p}.ca.op", dm, dm)]
        lns.append(LNSpec(f"{p}.ln2", dm))
        lins += [LinSpec(f"{p}.fc1", dm, arch.dec_ffn), LinSpec(f"{p}.fc2", arch.dec_ffn, dm)]
        lns.append(LNSpec(f"{p}.ln3", dm))
        lins += [LinSpec(f"dec.bbox.{i}.0", dm, dm), LinSpec(f"dec.bbox.{i}.1", dm, dm),
                 LinSpec(f"dec.bbox.{i}.2", dm, 4, "small")]
        lins.append(LinSpec(f"dec.cls.{i}", dm, arch.num_classes, "score"))
    return convs, lins, lns


def _gen(name: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def synth_weights(arch: Arch, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Deterministic re-conditioned random weights (SURVEY.md §8c "fixture recipe").

    Every tensor is drawn from its own generator keyed by (tensor name, seed), so the same
    weights are regenerated on the GPU box without HuggingFace or any checkpoint.
    BN: gamma=1, beta~N(0,.1), mean~N(0,.1), var~U(.5,1.5); convs He-normal; score heads
    W~N(0,.05) with bias -3 (a few dozen detections clear conf 0.25); bbox / attention-weight
    heads W~N(0,.02); sampling-offset bias ~N(0,1) so samples leave the map (zero padding taps).
    """
    convs, lins, lns = module_specs(arch)
    w: Dict[str, torch.Tensor] = {}

    def randn(name, shape, std):
        return torch.randn(shape, generator=_gen(name, seed), dtype=torch.float32) * std

    for c in convs:
        fan_in = c.cin * c.k * c.k
        w[c.name + ".conv.w"] = randn(c.name + ".conv.w", (c.cout, c.cin, c.k, c.k), c.gain * (2.0 / fan_in) ** 0.5)
        w[c.name + ".bn.g"] = torch.ones(c.cout)
        w[c.name + ".bn.b"] = randn(c.name + ".bn.b", (c.cout,), 0.1)
        w[c.name + ".bn.m"] = randn(c.name + ".bn.m", (c.cout,), 0.1)
        w[c.name + ".bn.v"] = torch.rand((c.cout,), generator=_gen(c.name + ".bn.v", seed)) + 0.5
    for l in lins:
        if l.kind == "score":
            std, bstd, bmean = 0.8 / l.cin ** 0.5, 0.0, -3.0   # = 0.05 at d_model 256
        elif l.kind == "small":
            std, bstd, bmean = 0.02, 0.02, 0.0
        elif l.kind == "offset":
            std, bstd, bmean = 0.02, 1.0, 0.0
        else:
            std, bstd, bmean = (1.0 / l.cin) ** 0.5, 0.02, 0.0
        w[l.name + ".w"] = randn(l.name + ".w", (l.cout, l.cin), std)
        w[l.name + ".b"] = randn(l.name + ".b", (l.cout,), bstd) + bmean
    for n in lns:
        w[n.name + ".g"] = 1.0 + randn(n.name + ".g", (n.dim,), 0.02)
        w[n.name + ".b"] = randn(n.name + ".b", (n.dim,), 0.02)
    return w


# --------------------------------------------------------------------------------------
# load-time folding (the build's `.deploy()`)
# --------------------------------------------------------------------------------------

def _fold_bn(w: Dict[str, torch.Tensor], name: str) -> Tuple[torch.Tensor, torch.Tensor]:
    """conv+BN -> (filter [Co,Ci,k,k], bias [Co]); frozen-BN affine of HF:...v2.py:740-751."""
    cw = w[name + ".conv.w"].double()
    g, b, m, v = (w[name + ".bn." + s].double() for s in "gbmv")
    scale = g / torch.sqrt(v + BN_EPS)
    return (cw * scale[:, None, None, None]), (b - m * scale)


def _ohwi(f: torch.Tensor) -> torch.Tensor:
    return f.permute(0, 2, 3, 1).contiguous()


def fold_weights(arch: Arch, w: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Un-fused parameters -> the fused fp32 tensors the HIP engine consumes.

    Naming of the output: "<layer>.w" is an OHWI filter [Co,kh,kw,Ci] (a linear is [Co,Ci]),
    "<layer>.b" its bias.  Folding is done in float64 and rounded once to fp32.
    """
    convs, lins, lns = module_specs(arch)
    spec = {c.name: c for c in convs}
    out: Dict[str, torch.Tensor] = {}

    def put(name, f, b):
        out[name + ".w"] = f.float().contiguous()
        out[name + ".b"] = b.float().contiguous()

    # stem conv 0: pad Ci 3 -> 8 so every filter row is a whole number of 16-byte chunks
    f, b = _fold_bn(w, "backbone.stem.0")
    f = _ohwi(f)
    f = torch.cat([f, torch.zeros(*f.shape[:3], 5, dtype=f.dtype)], dim=3)
    put("backbone.stem.0", f, b)
    for n in ("backbone.stem.1", "backbone.stem.2"):
        f, b = _fold_bn(w, n)
        put(n, _ohwi(f), b)
    for pfx, cin, cout, stride, first in backbone_blocks(arch):
        names = [".c1", ".c2", ".c3"] if arch.layer_type == "bottleneck" else [".c1", ".c2"]
        for s in names:
            f, b = _fold_bn(w, pfx + s)
            put(pfx + s, _ohwi(f), b)
        if (pfx + ".sc") in spec:
            f, b = _fold_bn(w, pfx + ".sc")
            f = _ohwi(f)                                   # [Co,1,1,Ci]; stride 2: the engine runs AvgPool2d(2,2) first
            put(pfx + ".sc", f, b)
    d, h = arch.enc_dim, arch.csp_hidden
    for l in range(3):
        f, b = _fold_bn(w, f"enc.proj.{l}")
        put(f"enc.proj.{l}", _ohwi(f), b)
    # AIFI: q|k share the (x+pos) input -> one GEMM with N=2d
    out["enc.aifi.qk.w"] = torch.cat([w["enc.aifi.q.w"], w["enc.aifi.k.w"]], 0).contiguous()
    out["enc.aifi.qk.b"] = torch.cat([w["enc.aifi.q.b"], w["enc.aifi.k.b"]], 0).contiguous()
    for n in ("v", "o", "fc1", "fc2"):
        out[f"enc.aifi.{n}.w"] = w[f"enc.aifi.{n}.w"].clone()
        out[f"enc.aifi.{n}.b"] = w[f"enc.aifi.{n}.b"].clone()
    for n in ("ln1", "ln2"):
        out[f"enc.aifi.{n}.g"] = w[f"enc.aifi.{n}.g"].clone()
        out[f"enc.aifi.{n}.b"] = w[f"enc.aifi.{n}.b"].clone()

    def csp(pfx):
        # conv1 | conv2 read the same fused map -> one 1x1 GEMM with N = 2h (HF:...v2.py:948-951)
        f1, b1 = _fold_bn(w, pfx + ".c1")
        f2, b2 = _fold_bn(w, pfx + ".c2")
        put(pfx + ".c12", _ohwi(torch.cat([f1, f2], 0)), torch.cat([b1, b2], 0))
        for j in range(3):
            # RepVGG re-parameterisation: 3x3 + (1x1 padded to the centre tap) (HF:...v2.py:907-923)
            f3, b3 = _fold_bn(w, f"{pfx}.rep{j}.k3")
            f1x, b1x = _fold_bn(w, f"{pfx}.rep{j}.k1")
            f3 = f3.clone()
            f3[:, :, 1, 1] += f1x[:, :, 0, 0]
            put(f"{pfx}.rep{j}", _ohwi(f3), b3 + b1x)
        if h != d:
            f, b = _fold_bn(w, pfx + ".c3")
            put(pfx + ".c3", _ohwi(f), b)

    for i in range(2):
        f, b = _fold_bn(w, f"enc.lat.{i}")
        put(f"enc.lat.{i}", _ohwi(f), b)
        csp(f"enc.fpn.{i}")
        f, b = _fold_bn(w, f"enc.down.{i}")
        put(f"enc.down.{i}", _ohwi(f), b)
        csp(f"enc.pan.{i}")
    for l in range(arch.n_levels):
        f, b = _fold_bn(w, f"dec.proj.{l}")
        put(f"dec.proj.{l}", _ohwi(f), b)
    for n in ("dec.enc_out.fc", "dec.enc_score", "dec.enc_bbox.0", "dec.enc_bbox.1", "dec.enc_bbox.2", "dec.qpos.1"):
        out[n + ".w"] = w[n + ".w"].clone()
        out[n + ".b"] = w[n + ".b"].clone()
    # qpos.0 takes the 4-vector reference box; the engine keeps boxes in 8-float rows
    q0 = w["dec.qpos.0.w"]
    out["dec.qpos.0.w"] = torch.cat([q0, torch.zeros(q0.shape[0], 4)], 1).contiguous()
    out["dec.qpos.0.b"] = w["dec.qpos.0.b"].clone()
    out["dec.enc_out.ln.g"] = w["dec.enc_out.ln.g"].clone()
    out["dec.enc_out.ln.b"] = w["dec.enc_out.ln.b"].clone()
    vps_w, vps_b = [], []
    for i in range(arch.dec_layers):
        p = f"dec.l{i}"
        out[p + ".sa.qk.w"] = torch.cat([w[p + ".sa.q.w"], w[p + ".sa.k.w"]], 0).contiguous()
        out[p + ".sa.qk.b"] = torch.cat([w[p + ".sa.q.b"], w[p + ".sa.k.b"]], 0).contiguous()
        for n in ("sa.v", "sa.o", "ca.op", "fc1", "fc2"):
            out[f"{p}.{n}.w"] = w[f"{p}.{n}.w"].clone()
            out[f"{p}.{n}.b"] = w[f"{p}.{n}.b"].clone()
        # sampling offsets | attention weights share (h+pos) -> one GEMM, N = 3*npts
        out[p + ".ca.offaw.w"] = torch.cat([w[p + ".ca.off.w"], w[p + ".ca.aw.w"]], 0).contiguous()
        out[p + ".ca.offaw.b"] = torch.cat([w[p + ".ca.off.b"], w[p + ".ca.aw.b"]], 0).contiguous()
        vps_w.append(w[p + ".ca.vp.w"])
        vps_b.append(w[p + ".ca.vp.b"])
        for n in ("ln1", "ln2", "ln3"):
            out[f"{p}.{n}.g"] = w[f"{p}.{n}.g"].clone()
            out[f"{p}.{n}.b"] = w[f"{p}.{n}.b"].clone()
        for j in range(3):
            out[f"dec.bbox.{i}.{j}.w"] = w[f"dec.bbox.{i}.{j}.w"].clone()
            out[f"dec.bbox.{i}.{j}.b"] = w[f"dec.bbox.{i}.{j}.b"].clone()
    # every decoder layer projects the SAME memory tokens (HF:...v2.py:177) -> one GEMM, N = L*d
    out["dec.vp_all.w"] = torch.cat(vps_w, 0).contiguous()
    out["dec.vp_all.b"] = torch.cat(vps_b, 0).contiguous()
    # only the last layer's logits reach the post-processor (HF:...v2.py:1880)
    last = arch.dec_layers - 1
    out["dec.cls.w"] = w[f"dec.cls.{last}.w"].clone()
    out["dec.cls.b"] = w[f"dec.cls.{last}.b"].clone()
    return out


# --------------------------------------------------------------------------------------
# flat blob:  "RTDW" u32 version u32 count | count x { u16 name_len, name, u32 ndim,
#             u32 dims[ndim], u64 offset, u64 nbytes } | 64-byte aligned fp32 payloads
# --------------------------------------------------------------------------------------
BLOB_MAGIC = b"RTDW"
BLOB_VERSION = 1


def pack_blob(tensors: Dict[str, torch.Tensor]) -> bytes:
    names = sorted(tensors)
    header = bytearray()
    header += BLOB_MAGIC + struct.pack("<II", BLOB_VERSION, len(names))
    entries = []
    for n in names:
        t = tensors[n]
        assert t.dtype == torch.float32, (n, t.dtype)
        entries.append((n.encode(), tuple(t.shape), t.numel() * 4))
    table_len = sum(2 + len(nb) + 4 + 4 * len(sh) + 16 for nb, sh, _ in entries)
    off = (len(header) + table_len + 63) // 64 * 64
    offs = []
    for _, _, nbytes in entries:
        offs.append(off)
        off = (off + nbytes + 63) // 64 * 64
    for (nb, sh, nbytes), o in zip(entries, offs):
        header += struct.pack("<H", len(nb)) + nb + struct.pack("<I", len(sh))
        header += struct.pack("<%dI" % len(sh), *sh) + struct.pack("<QQ", o, nbytes)
    buf = bytearray(off)
    buf[: len(header)] = header
    for n, o in zip(names, offs):
        a = tensors[n].contiguous().numpy().tobytes()
        buf[o : o + len(a)] = a
    return bytes(buf)


def unpack_blob(blob: bytes) -> Dict[str, np.ndarray]:
    assert blob[:4] == BLOB_MAGIC
    ver, count = struct.unpack_from("<II", blob, 4)
    assert ver == BLOB_VERSION
    p = 12
    out = {}
    for _ in range(count):
        (nl,) = struct.unpack_from("<H", blob, p); p += 2
        name = blob[p : p + nl].decode(); p += nl
        (nd,) = struct.unpack_from("<I", blob, p); p += 4
        dims = struct.unpack_from("<%dI" % nd, blob, p); p += 4 * nd
        off, nbytes = struct.unpack_from("<QQ", blob, p); p += 16
        out[name] = np.frombuffer(blob, dtype=np.float32, count=nbytes // 4, offset=off).reshape(dims)
    return out


def save_weights(path: str, arch: Arch, w: Dict[str, torch.Tensor]) -> None:
    """Write a model file this build's `RTDETRDetector.load_model` accepts (un-fused tensors)."""
    torch.save({"arch": arch.name, "model": w}, path)


# --------------------------------------------------------------------------------------
# Packed-blob cache.  One process per GPU means eight ranks of a node load the same checkpoint at the same moment (main.py:1236-1291: one
# engine per camera); each would read the file, fold BN / RepVGG and pack ~170 MB.  The packed blob is therefore kept on disk, keyed by
# what it was made from - variant, source (seed, or file path + size + mtime) and the text of this module (the folding rules) - written
# once with an atomic rename and read back by the other ranks.  RTD_BLOB_CACHE names the directory ("" or "0" switches the cache off).
# --------------------------------------------------------------------------------------
def _cache_dir():
    import os
    d = os.environ.get("RTD_BLOB_CACHE")
    if d is None:
        d = os.path.join(os.environ.get("XDG_CACHE_HOME", os.path.join(os.path.expanduser("~"), ".cache")), "telescope_cam_detection_amd", "blobs")
    return None if d in ("", "0") else d


def blob_cache_key(arch: Arch, source: str) -> str:
    import hashlib
    import os
    ident = source
    if not source.startswith("synthetic:") and os.path.exists(source):
        st = os.stat(source)
        ident = f"{os.path.abspath(source)}|{st.st_size}|{st.st_mtime_ns}"
    with open(os.path.abspath(__file__), "rb") as f:
        rules = hashlib.sha256(f.read()).hexdigest()[:16]
    return hashlib.sha256(f"{arch.name}|{ident}|{rules}|v{BLOB_VERSION}".encode()).hexdigest()[:32]


def cached_blob(arch: Arch, source: str, make_state) -> bytes:
    """The packed blob for (`arch`, `source`): from the cache when present and intact, else `pack_blob(fold_weights(arch, make_state()))`,
    stored for the next process.  Any cache trouble (read-only home, a torn file) falls back to building it: the cache is an optimisation."""
    import os
    import tempfile
    d = _cache_dir()
    path = os.path.join(d, blob_cache_key(arch, source) + ".rtdw") if d else None
    if path and os.path.exists(path):
        try:
            with open(path, "rb") as f:
                blob = f.read()
            if blob[:4] == BLOB_MAGIC and len(blob) >= 12:
                unpack_blob(blob)      # validates the table against the file's length (raises on a torn file)
                return blob
        except Exception:
            pass
    blob = pack_blob(fold_weights(arch, make_state()))
    if path:
        try:
            os.makedirs(d, exist_ok=True)
            fd, tmp = tempfile.mkstemp(dir=d, suffix=".tmp")
            with os.fdopen(fd, "wb") as f:
                f.write(blob)
            os.replace(tmp, path)                                    # atomic: a concurrent reader sees the old state or the whole file
        except OSError:
            pass
    return blob
