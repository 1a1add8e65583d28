"""CPU ORACLE - test infrastructure only, never part of the product path.

A plain fp32 PyTorch-on-CPU restatement of the reference's RT-DETRv2 detection path
(`RTDETRDetector.preprocess/detect/detect_batch`, /root/reference/src/rtdetr_detector.py:206-403).
The network arithmetic itself is NOT in the reference tree: it lives in the un-vendored,
unpinned third-party repo lyuwenyu/RT-DETR (`rtdetrv2_pytorch`, requirements.txt:6-8 of the
reference is only a comment; call sites src/rtdetr_detector.py:102,132,164-170).  This file
restates that published algorithm following HuggingFace transformers 5.15.0's line-for-line
port of it ("HF:" citations = transformers/models/...), which is present in the build container
and is used by oracle/make_golden.py to PIN this restatement (tests/golden/*.npz).

Pinning status: the reference holds no golden vector / known-answer test for this path
 - parity is pinned against HF run in-container with seeded weights, i.e.
"pinned to the third-party port", and "parity unpinned" with respect to the reference's own tests.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It consumes the UN-fused parameters of telescope_cam_detection_amd.weights.module_specs().
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
LN_EPS = 1e-5

COCO_CLASSES = [
    "person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light",
    "fire hydrant", "stop sign", "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep", "cow",
    "elephant", "bear", "zebra", "giraffe", "backpack", "umbrella", "handbag", "tie", "suitcase", "frisbee",
    "skis", "snowboard", "sports ball", "kite", "baseball bat", "baseball glove", "skateboard", "surfboard",
    "tennis racket", "bottle", "wine glass", "cup", "fork", "knife", "spoon", "bowl", "banana", "apple",
    "sandwich", "orange", "broccoli", "carrot", "hot dog", "pizza", "donut", "cake", "chair", "couch",
    "potted plant", "bed", "dining table", "toilet", "tv", "laptop", "mouse", "remote", "keyboard", "cell phone",
    "microwave", "oven", "toaster", "sink", "refrigerator", "book", "clock", "vase", "scissors", "teddy bear",
    "hair drier", "toothbrush",
]  # src/coco_constants.py:7-17
WILDLIFE_CLASSES = {0, 14, 15, 16, 21}  # src/coco_constants.py:23-29


# ----------------------------------------------------------------------------- preprocess
def preprocess(frame_bgr: np.ndarray, input_size: Tuple[int, int]) -> Tuple[torch.Tensor, Tuple[int, int]]:
    """src/rtdetr_detector.py:206-236: BGR->RGB, ToPILImage, Resize(input_size) (PIL bilinear,
    antialiased, aspect-distorting), ToTensor (/255, CHW).  Returns ([1,3,H,W] fp32, (w, h))."""
    from PIL import Image

    orig_h, orig_w = frame_bgr.shape[:2]
    rgb = np.ascontiguousarray(frame_bgr[:, :, ::-1])
    img = Image.fromarray(rgb)                       # ToPILImage on an HWC uint8 ndarray
    th, tw = input_size
    if (orig_h, orig_w) != (th, tw):
        img = img.resize((tw, th), Image.BILINEAR)   # T.Resize((h,w)) -> PIL resize((w,h), BILINEAR)
    arr = np.asarray(img, dtype=np.uint8)
    x = torch.from_numpy(arr.copy()).permute(2, 0, 1).float().div(255.0)[None]   # ToTensor
    return x, (orig_w, orig_h)


# ----------------------------------------------------------------------------- building blocks
def conv_bn(w, name, x, stride=1, padding=None, act=None):
    """conv (no bias) + frozen BN + activation.  HF:rt_detr_resnet.py:37-68, HF:v2.py:722-758,817-835."""
    cw = w[name + ".conv.w"]
    k = cw.shape[-1]
    pad = (k // 2) if padding is None else padding
    y = F.conv2d(x, cw, None, stride=stride, padding=pad)
    g, b, m, v = (w[name + ".bn." + s] for s in "gbmv")
    scale = g * torch.rsqrt(v + BN_EPS)
    y = y * scale.view(1, -1, 1, 1) + (b - m * scale).view(1, -1, 1, 1)
    if act == "relu":
        y = F.relu(y)
    elif act == "silu":
        y = F.silu(y)
    return y


def linear(w, name, x):
    return F.linear(x, w[name + ".w"], w[name + ".b"])


def layer_norm(w, name, x):
    return F.layer_norm(x, (x.shape[-1],), w[name + ".g"], w[name + ".b"], LN_EPS)


def backbone(arch, w, x):
    """PResNet-vd.  HF:rt_detr_resnet.py:71-310; returns the stage 2,3,4 maps (out_indices [2,3,4])."""
    from telescope_cam_detection_amd.weights import backbone_blocks, block_has_shortcut

    x = conv_bn(w, "backbone.stem.0", x, stride=2, act="relu")
    x = conv_bn(w, "backbone.stem.1", x, act="relu")
    x = conv_bn(w, "backbone.stem.2", x, act="relu")
    x = F.max_pool2d(x, 3, 2, 1)
    feats = {}
    for pfx, cin, cout, stride, first in backbone_blocks(arch):
        res = x
        if arch.layer_type == "bottleneck":
            y = conv_bn(w, pfx + ".c1", x, act="relu")
            y = conv_bn(w, pfx + ".c2", y, stride=stride, act="relu")
            y = conv_bn(w, pfx + ".c3", y)
        else:
            y = conv_bn(w, pfx + ".c1", x, stride=stride, act="relu")
            y = conv_bn(w, pfx + ".c2", y)
        if block_has_shortcut(arch, cin, cout, stride, first):
            if stride == 2:
                res = F.avg_pool2d(res, 2, 2, 0, ceil_mode=True)     # resnet.py:199-205 / 150-157
            res = conv_bn(w, pfx + ".sc", res)
        x = F.relu(y + res)
        feats[pfx] = x
    outs = []
    for si in (1, 2, 3):
        outs.append(feats[f"backbone.s{si}.b{arch.depths[si] - 1}"])
    return outs


def sincos_pos(h: int, w_: int, dim: int, temperature: float = 10000.0) -> torch.Tensor:
    """HF:v2.py:955-1000 : [sin_h | cos_h | sin_w | cos_w], float64 arithmetic, -> fp32 [h*w, dim]."""
    pos_dim = dim // 4
    omega = torch.arange(pos_dim, dtype=torch.float64) / pos_dim
    omega = 1.0 / temperature ** omega
    gh, gw = torch.meshgrid(torch.arange(h, dtype=torch.float64), torch.arange(w_, dtype=torch.float64), indexing="ij")
    eh = gh.flatten().outer(omega)
    ew = gw.flatten().outer(omega)
    return torch.cat([eh.sin(), eh.cos(), ew.sin(), ew.cos()], dim=1).float()


def mha(w, pfx, x, pos, heads, fused_names=("q", "k", "v", "o")):
    """HF:v2.py:273-336 (eager): q,k from x+pos; v from x; softmax(q k^T / sqrt(d)) v; o_proj."""
    B, L, D = x.shape
    d = D // heads
    qk_in = x + pos if pos is not None else x
    q = linear(w, f"{pfx}.q", qk_in).view(B, L, heads, d).transpose(1, 2)
    k = linear(w, f"{pfx}.k", qk_in).view(B, L, heads, d).transpose(1, 2)
    v = linear(w, f"{pfx}.v", x).view(B, L, heads, d).transpose(1, 2)
    a = torch.matmul(q, k.transpose(2, 3)) * (d ** -0.5)
    a = F.softmax(a, dim=-1)
    o = torch.matmul(a, v).transpose(1, 2).reshape(B, L, D)
    return linear(w, f"{pfx}.o", o)


def aifi(arch, w, x):
    """HF:v2.py:1041-1095 + layer :838-904 (post-norm, GELU)."""
    B, C, H, W = x.shape
    t = x.flatten(2).transpose(1, 2)
    pos = sincos_pos(H, W, C)[None]
    t = layer_norm(w, "enc.aifi.ln1", t + mha(w, "enc.aifi", t, pos, arch.enc_heads))
    y = linear(w, "enc.aifi.fc2", F.gelu(linear(w, "enc.aifi.fc1", t)))
    t = layer_norm(w, "enc.aifi.ln2", t + y)
    return t.transpose(1, 2).reshape(B, C, H, W).contiguous()


def csp_rep(arch, w, pfx, x):
    """HF:v2.py:926-952 ; RepVGG block :907-923 (un-fused: 3x3 branch + 1x1 branch, then SiLU)."""
    h1 = conv_bn(w, pfx + ".c1", x, act="silu")
    for j in range(3):
        h1 = F.silu(conv_bn(w, f"{pfx}.rep{j}.k3", h1, padding=1) + conv_bn(w, f"{pfx}.rep{j}.k1", h1, padding=0))
    h2 = conv_bn(w, pfx + ".c2", x, act="silu")
    y = h1 + h2
    if (pfx + ".c3.conv.w") in w:
        y = conv_bn(w, pfx + ".c3", y, act="silu")
    return y


def hybrid_encoder(arch, w, feats):
    """HF:v2.py:1348-1360,1512 (input proj) ; :1183-1209 (AIFI, FPN top-down, PAN bottom-up)."""
    f = [conv_bn(w, f"enc.proj.{l}", feats[l]) for l in range(3)]
    f[2] = aifi(arch, w, f[2])
    fpn = [f[2]]
    for idx in range(2):
        backbone_map = f[1 - idx]
        top = conv_bn(w, f"enc.lat.{idx}", fpn[-1], act="silu")
        fpn[-1] = top
        up = F.interpolate(top, scale_factor=2.0, mode="nearest")
        fpn.append(csp_rep(arch, w, f"enc.fpn.{idx}", torch.cat([up, backbone_map], dim=1)))
    fpn.reverse()
    pan = [fpn[0]]
    for idx in range(2):
        down = conv_bn(w, f"enc.down.{idx}", pan[-1], stride=2, act="silu")
        pan.append(csp_rep(arch, w, f"enc.pan.{idx}", torch.cat([down, fpn[idx + 1]], dim=1)))
    return pan


def anchors_and_mask(shapes: Sequence[Tuple[int, int]], grid_size: float = 0.05):
    """HF:v2.py:1423-1449 ; fp32 arithmetic; invalid anchors become finfo.max (upstream: inf)."""
    out = []
    for lvl, (h, w_) in enumerate(shapes):
        gy, gx = torch.meshgrid(torch.arange(h).float(), torch.arange(w_).float(), indexing="ij")
        xy = torch.stack([gx, gy], -1).unsqueeze(0) + 0.5
        xy[..., 0] /= w_
        xy[..., 1] /= h
        wh = torch.ones_like(xy) * grid_size * (2.0 ** lvl)
        out.append(torch.cat([xy, wh], -1).reshape(-1, h * w_, 4))
    a = torch.cat(out, 1)
    valid = ((a > 1e-2) * (a < 1 - 1e-2)).all(-1, keepdim=True)
    a = torch.log(a / (1 - a))
    a = torch.where(valid, a, torch.full((), torch.finfo(torch.float32).max))
    return a, valid


def inverse_sigmoid(x, eps=1e-5):
    """HF:v2.py:548-552"""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def mlp_head(w, pfx, x, n):
    """HF:v2.py:1659-1675 : ReLU between layers, none after the last."""
    for i in range(n):
        x = linear(w, f"{pfx}.{i}", x)
        if i < n - 1:
            x = F.relu(x)
    return x


def ms_deform_attn(arch, w, pfx, hs, pos, ref, memory, shapes):
    """HF:v2.py:119-225 (module) + :44-115 (sampler, method='default').

    ref: [B,Q,4] sigmoid boxes.  loc = ref_xy + off * (1/n_points) * ref_wh * offset_scale ;
    grid_sample(bilinear, zeros, align_corners=False) on each level ; softmax over levels*points.
    """
    B, Q, D = hs.shape
    H, Lv, P = arch.dec_heads, arch.n_levels, arch.n_points
    d = D // H
    S = memory.shape[1]
    q = hs + pos
    value = linear(w, pfx + ".vp", memory).view(B, S, H, d)
    off = linear(w, pfx + ".off", q).view(B, Q, H, Lv * P, 2)
    aw = F.softmax(linear(w, pfx + ".aw", q).view(B, Q, H, Lv * P), -1)
    scale = torch.full((Lv * P,), 1.0 / P).unsqueeze(-1)
    r = ref[:, :, None, :]                                             # [B,Q,1,4]
    offset = off * scale * r[:, :, None, :, 2:] * arch.offset_scale
    loc = r[:, :, None, :, :2] + offset                                # [B,Q,H,LvP,2]
    grids = (2 * loc - 1).permute(0, 2, 1, 3, 4).flatten(0, 1)         # [B*H,Q,LvP,2]
    vlist = value.permute(0, 2, 3, 1).flatten(0, 1).split([h * w_ for h, w_ in shapes], dim=-1)
    sampled = []
    for lvl, (h, w_) in enumerate(shapes):
        v = vlist[lvl].reshape(B * H, d, h, w_)
        g = grids[:, :, lvl * P:(lvl + 1) * P]
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    a = aw.permute(0, 2, 1, 3).reshape(B * H, 1, Q, Lv * P)
    o = (torch.cat(sampled, dim=-1) * a).sum(-1).view(B, H * d, Q).transpose(1, 2).contiguous()
    return linear(w, pfx + ".op", o)


def decoder_and_heads(arch, w, pan, collect=None, force_topk=None):
    """HF:v2.py:1533-1623 (query selection) ; :555-661 (decoder loop) ; :1880-1881 (last layer).
    force_topk [B,Q] (tests only): run the decoder on these memory tokens instead of the restatement's own top-Q - the reference under the
    selection of the implementation being checked, for frames where a near-tie at the rank-Q cut fell the other way."""
    srcs, shapes = [], []
    for l, fmap in enumerate(pan):
        s = conv_bn(w, f"dec.proj.{l}", fmap)
        shapes.append(tuple(s.shape[-2:]))
        srcs.append(s.flatten(2).transpose(1, 2))
    src = torch.cat(srcs, 1)                                            # [B,S,D]
    anchors, valid = anchors_and_mask(shapes)
    memory = valid.to(src.dtype) * src
    om = layer_norm(w, "dec.enc_out.ln", linear(w, "dec.enc_out.fc", memory))
    enc_cls = linear(w, "dec.enc_score", om)
    enc_box = mlp_head(w, "dec.enc_bbox", om, 3) + anchors
    _, topk = torch.topk(enc_cls.max(-1).values, arch.num_queries, dim=1)
    if force_topk is not None:
        topk = torch.as_tensor(force_topk, dtype=torch.int64).view(topk.shape)
    ref_unact = enc_box.gather(1, topk.unsqueeze(-1).repeat(1, 1, 4))
    target = om.gather(1, topk.unsqueeze(-1).repeat(1, 1, om.shape[-1]))
    if collect is not None:
        collect.update(memory_tokens=src, enc_cls_max=enc_cls.max(-1).values, topk=topk,
                       ref_unact=ref_unact, target=target)
    hs = target
    ref = torch.sigmoid(ref_unact)
    logits = None
    for i in range(arch.dec_layers):
        p = f"dec.l{i}"
        pos = mlp_head(w, "dec.qpos", ref, 2)
        hs = layer_norm(w, p + ".ln1", hs + mha(w, p + ".sa", hs, pos, arch.dec_heads))
        hs = layer_norm(w, p + ".ln2", hs + ms_deform_attn(arch, w, p + ".ca", hs, pos, ref, src, shapes))
        hs = layer_norm(w, p + ".ln3", hs + linear(w, p + ".fc2", F.relu(linear(w, p + ".fc1", hs))))
        ref = torch.sigmoid(mlp_head(w, f"dec.bbox.{i}", hs, 3) + inverse_sigmoid(ref))
        logits = linear(w, f"dec.cls.{i}", hs)
        if collect is not None:
            collect[f"dec{i}.hs"] = hs
            collect[f"dec{i}.ref"] = ref
    return logits, ref


def postprocess(logits, boxes, orig_sizes_wh):
    """Upstream RTDETRPostProcessor (deploy mode) == HF:rt_detr/image_processing_rt_detr.py:510-533.

    cxcywh->xyxy, x (w,h,w,h) of the ORIGINAL frame, sigmoid, top-Q over Q*C, label = idx % C.
    Returns (labels int64 [B,Q], boxes [B,Q,4], scores [B,Q]) - what src/rtdetr_detector.py:257 unpacks.
    """
    B, Q, C = logits.shape
    cx, cy, w_, h = boxes.unbind(-1)
    xyxy = torch.stack([cx - 0.5 * w_, cy - 0.5 * h, cx + 0.5 * w_, cy + 0.5 * h], -1)
    wh = torch.as_tensor(orig_sizes_wh, dtype=torch.float32).view(B, 2)
    xyxy = xyxy * wh.repeat(1, 2)[:, None, :]
    scores = torch.sigmoid(logits)
    scores, index = torch.topk(scores.flatten(1), Q, dim=-1)
    labels = index % C
    q = index // C
    out_boxes = xyxy.gather(1, q.unsqueeze(-1).repeat(1, 1, 4))
    return labels, out_boxes, scores


@torch.no_grad()
def model_forward(arch, w, images, orig_sizes_wh, collect=None, force_topk=None):
    """`Model.forward` of src/rtdetr_detector.py:161-170: network + post-processor."""
    feats = backbone(arch, w, images)
    if collect is not None:
        for i, f in enumerate(feats):
            collect[f"backbone{i}"] = f
    pan = hybrid_encoder(arch, w, feats)
    if collect is not None:
        for i, f in enumerate(pan):
            collect[f"enc{i}"] = f
    logits, boxes = decoder_and_heads(arch, w, pan, collect, force_topk)
    if collect is not None:
        collect["logits"] = logits
        collect["pred_boxes"] = boxes
    return postprocess(logits, boxes, orig_sizes_wh)


def format_detections(labels, boxes, scores, conf_threshold=0.25, wildlife_only=True) -> List[dict]:
    """Restatement of the per-row loop of src/rtdetr_detector.py:262-303 for ONE frame."""
    labels = np.asarray(labels)
    boxes = np.asarray(boxes)
    scores = np.asarray(scores)
    dets = []
    for i in range(len(labels)):
        score = float(scores[i])
        if score < conf_threshold:
            continue
        class_id = int(labels[i])
        if wildlife_only and class_id not in WILDLIFE_CLASSES:
            continue
        x1, y1, x2, y2 = (float(v) for v in boxes[i])
        name = COCO_CLASSES[class_id] if class_id < len(COCO_CLASSES) else f"class_{class_id}"
        dets.append({"class_id": class_id, "class_name": name, "confidence": score,
                     "bbox": {"x1": x1, "y1": y1, "x2": x2, "y2": y2, "area": int((x2 - x1) * (y2 - y1))}})
    return dets


@torch.no_grad()
def detect_batch(arch, w, frames_bgr, input_size, conf_threshold=0.25, wildlife_only=True):
    """src/rtdetr_detector.py:307-403 end to end on the CPU."""
    xs, sizes = [], []
    for f in frames_bgr:
        x, wh = preprocess(f, input_size)
        xs.append(x)
        sizes.append(wh)
    labels, boxes, scores = model_forward(arch, w, torch.cat(xs, 0), sizes)
    return [format_detections(labels[i].numpy(), boxes[i].numpy(), scores[i].numpy(), conf_threshold, wildlife_only)
            for i in range(len(frames_bgr))]
