"""Generate tests/golden/*.npz by running HuggingFace transformers' RT-DETRv2 IN THIS CONTAINER.

Run from the repo root:   python oracle/make_golden.py [case ...]

The reference's RT-DETR arithmetic is the absent third-party repo lyuwenyu/RT-DETR
(src/rtdetr_detector.py:23,73-76); HF transformers 5.15.0 ships a line-for-line port of it and
is importable here (never on the GPU box).  This script

  1. regenerates the seeded synthetic weights (telescope_cam_detection_amd.weights.synth_weights),
  2. loads them into `RTDetrV2ForObjectDetection` built FROM A CONFIG OBJECT (no hub access),
  3. runs HF's forward + `post_process_object_detection` semantics on seeded uint8 frames
     pre-processed exactly as src/rtdetr_detector.py:206-236 does (PIL), and
  4. stores the inputs' seeds, the full final (labels, boxes, scores), the pre-top-k heads and
     strided samples of intermediate tensors.

tests/test_oracle_golden.py then pins oracle/rtdetr_oracle.py against these files; the GPU
parity tests pin the HIP path against the oracle and against the same files.
Only data is stored - no reference or HF source text.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from telescope_cam_detection_amd.arch import ARCHS  # noqa: E402
from telescope_cam_detection_amd.synth import make_frame  # noqa: E402
from telescope_cam_detection_amd.weights import backbone_blocks, block_has_shortcut, module_specs, synth_weights  # noqa: E402
from oracle import rtdetr_oracle as orc  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# name -> (arch, weight seed, input_size (H,W), list of (frame seed, frame H, frame W), frame kind)
# kind "noise" = the reference benchmark's uniform-noise frames, "scene" = structured synthetic frames
CASES = {
    # BASELINE config 1: R18 640x640 bs1 (+ the 1280x720 / 1920x1080 frames of the reference's
    # tests/test_inference.py:64-68 through the non-identity resize)
    "c1_r18_640_bs1": ("r18", 0, (640, 640), [(1000, 640, 640)], "noise"),
    "c1_r18_640_scene": ("r18", 0, (640, 640), [(1100, 640, 640)], "scene"),
    "c1_r18_640_resize": ("r18", 0, (640, 640), [(1001, 720, 1280), (1002, 1080, 1920)], "noise"),
    # BASELINE config 2: R50 640x640 bs8
    "c2_r50_640_bs8": ("r50", 0, (640, 640), [(2000 + i, 640, 640) for i in range(8)], "noise"),
    "c2_r50_640_scene_bs2": ("r50", 0, (640, 640), [(2100, 640, 640), (2101, 480, 704)], "scene"),
    # BASELINE config 3 (bs1 slice of it, structured frames)
    "c3_r101_1280_bs1": ("r101", 0, (1280, 1280), [(3000, 1280, 1280)], "scene"),
    # BASELINE config 3 at its full batch: 4 of the reference benchmark's noise frames (seeds 3000 + i)
    "c3_r101_1280_bs4": ("r101", 0, (1280, 1280), [(3000 + i, 1280, 1280) for i in range(4)], "noise"),
    # the reference's largest advertised input (config/config.yaml:122: 1920x1920, "~150-250 ms/frame"): a 1080p camera frame stretched to
    # 1920 x 1920 by the resampler, the reference's default model (R18); 1e-2 px is 5.2e-6 of this frame, 75 600 memory tokens
    "c4_r18_1920_bs1": ("r18", 0, (1920, 1920), [(5000, 1080, 1920)], "scene"),
    # small graph-shaped cases for fast unit tests (the network itself only accepts input sizes
    # that are multiples of 32: the FPN concat of a 2x-upsampled map fails otherwise, in HF and
    # upstream alike); non-square maps and resized frames included
    "t_tiny_160": ("tiny", 1, (160, 160), [(4000, 160, 160), (4001, 160, 160)], "scene"),
    "t_tiny_160x224": ("tiny", 1, (160, 224), [(4002, 160, 224)], "noise"),
    "t_tinyb_192x128": ("tinyb", 2, (192, 128), [(4003, 192, 128), (4004, 100, 90), (4005, 300, 260)], "scene"),
    # every width a multiple of 32: the small case of the f16x3 engine (non-square, one resized frame)
    "t_tinyc_160x224": ("tinyc", 3, (160, 224), [(4006, 160, 224), (4007, 200, 150)], "scene"),
}


from telescope_cam_detection_amd.checkpoint import hf_key_map  # noqa: E402  (mine -> HF names; one table for golden generation and the converter)


def build_hf(arch, w):
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
    model = RTDetrV2ForObjectDetection(cfg).eval()
    km = hf_key_map(arch)
    sd = {hf: w[mine] for mine, hf in km.items()}
    # the decoder holds aliases of the heads (model.decoder.class_embed == class_embed)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    ok_missing = ("model.decoder.class_embed", "model.decoder.bbox_embed", "model.denoising_class_embed",
                  "n_points_scale", "num_batches_tracked")
    bad = [k for k in missing if not any(s in k for s in ok_missing)]
    assert not bad and not unexpected, (bad[:10], unexpected[:10])
    assert set(km) == set(w), sorted(set(w) ^ set(km))[:10]
    # heads must be independent per layer and visible through the decoder aliases
    assert model.model.decoder.bbox_embed[1].layers[0].weight.data_ptr() == model.bbox_embed[1].layers[0].weight.data_ptr()
    if arch.dec_layers > 1:
        assert model.bbox_embed[0].layers[0].weight.data_ptr() != model.bbox_embed[1].layers[0].weight.data_ptr()
        assert torch.equal(model.class_embed[1].weight, w["dec.cls.1.w"])
    return model




def sample(t: torch.Tensor, n=2048):
    f = t.detach().float().flatten()
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().copy()


@torch.no_grad()
def run_case(name):
    arch_name, wseed, input_size, frames, kind = CASES[name]
    arch = ARCHS[arch_name]
    w = synth_weights(arch, wseed)
    model = build_hf(arch, w)
    xs, sizes = [], []
    for fs, fh, fw in frames:
        x, wh = orc.preprocess(make_frame(kind, fs, fh, fw), input_size)
        xs.append(x)
        sizes.append(wh)
    x = torch.cat(xs, 0)
    t0 = time.time()
    out = model(pixel_values=x)
    t_hf = time.time() - t0
    feats = model.model.backbone.model(x).feature_maps
    g = {}
    # final outputs through the post-processor semantics (HF:image_processing_rt_detr.py:510-533)
    labels, boxes, scores = orc.postprocess(out.logits, out.pred_boxes, sizes)
    from transformers.models.rt_detr.image_processing_pil_rt_detr import RTDetrImageProcessorPil  # torchvision-free twin
    res = RTDetrImageProcessorPil().post_process_object_detection(
        out, threshold=-1.0, target_sizes=[(h, w_) for (w_, h) in sizes])
    for i, r in enumerate(res):   # the HF processor and the restated post-processor must agree exactly
        assert torch.equal(r["scores"], scores[i]) and torch.equal(r["labels"], labels[i]) and torch.equal(r["boxes"], boxes[i])
    g["labels"] = labels.numpy().astype(np.int32)
    g["boxes"] = boxes.numpy()
    g["scores"] = scores.numpy()
    g["logits"] = out.logits.numpy()
    g["pred_boxes"] = out.pred_boxes.numpy()
    g["enc_cls_max"] = out.enc_outputs_class.max(-1).values.numpy()
    hf_topk = torch.topk(out.enc_outputs_class.max(-1).values, arch.num_queries, dim=1)[1]
    g["topk"] = hf_topk.numpy().astype(np.int32)      # query slot -> memory token
    g["enc_topk_bboxes"] = out.enc_topk_bboxes.numpy()
    g["init_ref_unact"] = out.init_reference_points.numpy()
    for i, f in enumerate(feats):
        g[f"s_backbone{i}"] = sample(f)
        g[f"shape_backbone{i}"] = np.array(f.shape)
    for i, f in enumerate(out.encoder_last_hidden_state):
        g[f"s_enc{i}"] = sample(f)
    for i in range(arch.dec_layers):
        g[f"s_dec{i}.hs"] = sample(out.intermediate_hidden_states[:, i])
        g[f"dec{i}.ref"] = out.intermediate_reference_points[:, i].numpy()
    g["input_sample"] = sample(x)
    g["meta"] = np.array([wseed, input_size[0], input_size[1], len(frames)], dtype=np.int64)
    g["frames"] = np.array(frames, dtype=np.int64)
    g["kind"] = np.array(kind)
    # oracle check right here, so a fixture is never written for a restatement that disagrees
    t0 = time.time()
    col = {}
    ol, ob, os_ = orc.model_forward(arch, w, x, sizes, collect=col)
    t_or = time.time() - t0
    # query ORDER is decided by near-ties of the encoder scores; compare per selected token
    so, sh = torch.sort(col["topk"], 1), torch.sort(hf_topk, 1)
    same_topk = torch.equal(so.values, sh.values)
    gat = lambda t, idx: t.gather(1, idx.unsqueeze(-1).expand(-1, -1, t.shape[-1]))
    d_logit = (gat(col["logits"], so.indices) - gat(out.logits, sh.indices)).abs().max().item()
    d_box = (gat(col["pred_boxes"], so.indices) - gat(out.pred_boxes, sh.indices)).abs().max().item()
    d_index = (col["logits"] - out.logits).abs().max().item()
    gaps = torch.sort(out.enc_outputs_class.max(-1).values, 1, descending=True).values
    gap_k = (gaps[:, arch.num_queries - 1] - gaps[:, arch.num_queries]).min().item()
    print(f"[{name}] HF {t_hf:.2f}s oracle {t_or:.2f}s  max|dlogit|={d_logit:.3e} max|dbox|={d_box:.3e} "
          f"(index-wise {d_index:.1e}) same_topk_set={same_topk} gap@k={gap_k:.2e}  score>0.25: {(scores > 0.25).sum().item()}  "
          f"score range [{scores.min().item():.3f},{scores.max().item():.3f}]")
    # a fixture is only written when the restatement agrees with HF on it (asserted, not just printed): per selected
    # token 1e-4 on logits and 1e-5 on normalised boxes (the bounds tests/test_oracle_golden.py applies), and the selected token SETS
    # equal unless the rank-Q / rank-Q+1 scores are a rounding-level near-tie
    assert d_logit <= 1e-4 and d_box <= 1e-5, (name, d_logit, d_box)
    assert same_topk or gap_k <= 2e-5, (name, same_topk, gap_k)
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    np.savez_compressed(os.path.join(GOLDEN_DIR, name + ".npz"), **g)


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    names = sys.argv[1:] or list(CASES)
    for n in names:
        run_case(n)
