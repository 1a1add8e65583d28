"""GPU: end-to-end parity of the HIP path (through the C ABI) against the CPU oracle and the
committed HF-generated golden fixtures, stage by stage and on the final (labels, boxes, scores).

Tolerances.  BASELINE.json's north star: |dscore| <= 1e-3, |dbox| <= 1e-2 px.  The fp32 engine
(v_mfma_f32_32x32x2_f32 everywhere) is held to exactly that.  The bf16 engine (bf16 storage + MFMA,
fp32 accumulate, fp32 selection/decoder) cannot meet 1e-2 px in general - a bf16 activation carries
2^-9 relative rounding and both top-k selections are discontinuous - so it is held to stage-level
relative errors and, with the query selection forced to the oracle's, to 2e-2 / 2 px; the measured
figures are printed and recorded in DESIGN.md.
"""
import numpy as np
import pytest
import torch

from oracle import rtdetr_oracle as orc
from tests.util import load_case, match_detections, weights_for

pytestmark = pytest.mark.gpu


def make_engine(arch, w, frames, input_size, precision, use_graph=False, profile=0):
    from telescope_cam_detection_amd import _capi
    from telescope_cam_detection_amd.weights import fold_weights, pack_blob
    blob = pack_blob(fold_weights(arch, w))
    prec = _capi.precision_code(precision)
    return _capi.Engine(arch, blob, device=0, precision=prec, max_batch=len(frames), input_size=input_size, use_graph=use_graph,
                        profile=profile)


def oracle_run(arch, w, frames, input_size):
    torch.set_num_threads(min(16, len(__import__('os').sched_getaffinity(0))))
    xs, sizes = zip(*[orc.preprocess(f, input_size) for f in frames])
    col = {}
    out = orc.model_forward(arch, w, torch.cat(xs, 0), list(sizes), collect=col)
    col["input"] = torch.cat(xs, 0)
    return out, col


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def nchw(t):
    return np.transpose(t, (0, 3, 1, 2))


FP32_CASES = ["t_tiny_160", "t_tiny_160x224", "t_tinyb_192x128", "t_tinyc_160x224", "c1_r18_640_bs1", "c1_r18_640_scene", "c1_r18_640_resize",
              "c2_r50_640_scene_bs2"]


@pytest.mark.parametrize("name", FP32_CASES)
def test_fp32_engine_matches_oracle_and_golden(name):
    arch, wseed, input_size, frames, g = load_case(name)
    w = weights_for(arch, wseed)
    (ol, ob, osc), col = oracle_run(arch, w, frames, input_size)
    eng = make_engine(arch, w, frames, input_size, "fp32")
    labels, boxes, scores = eng.infer_raw(frames)
    # stage 0: preprocess is integer/byte work -> bit-exact
    x = nchw(eng.debug_tensor("input"))[:, :3]
    np.testing.assert_array_equal(x, col["input"].numpy())
    # conv trunk + encoder
    for i in range(3):
        e = rel_err(nchw(eng.debug_tensor(f"backbone{i}")), col[f"backbone{i}"].numpy())
        assert e < 2e-5, (f"backbone{i}", e)
    for i in range(3):
        e = rel_err(nchw(eng.debug_tensor(f"enc{i}")), col[f"enc{i}"].numpy())
        assert e < 5e-5, (f"enc{i}", e)
    mx = eng.debug_tensor("enc_cls_max")[:, :, 0, 0]
    np.testing.assert_allclose(mx, col["enc_cls_max"].numpy(), atol=2e-4)
    np.testing.assert_allclose(mx, g["enc_cls_max"], atol=2e-4)
    # final outputs vs oracle AND vs the HF fixture: 1e-3 on scores, 1e-2 px on boxes, order-tolerant
    for b in range(len(frames)):
        for rl, rb, rs in ((ol[b].numpy(), ob[b].numpy(), osc[b].numpy()), (g["labels"][b], g["boxes"][b], g["scores"][b])):
            m, n, ws, wb = match_detections(rl, rb, rs, labels[b], boxes[b], scores[b], 1e-3, 1e-2)
            print(f"{name}[{b}] matched {m}/{n} worst dscore={ws:.2e} dbox={wb:.2e}px")
            # measured: every row on every case; ONE row is the allowance for a near-tie at a top-k cut (gap ~1e-5, see make_golden output)
            assert m >= n - 1, (m, n, ws, wb)
        assert (np.diff(scores[b]) <= 0).all(), "scores must be descending"
    eng.close()


@pytest.mark.parametrize("name", ["c2_r50_640_bs8", "c3_r101_1280_bs1", "c3_r101_1280_bs4"])
def test_fp32_engine_full_size_configs_against_hf_fixtures(name):
    """BASELINE configs 2 (R50 640 bs8) and 3 (R101 1280) at full size, against the committed HF outputs only
    (the CPU oracle would need ~10 s per case on the box)."""
    arch, wseed, input_size, frames, g = load_case(name)
    w = weights_for(arch, wseed)
    eng = make_engine(arch, w, frames, input_size, "fp32", use_graph=True)
    labels, boxes, scores = eng.infer_raw(frames)
    mx = eng.debug_tensor("enc_cls_max")[:, :, 0, 0]
    np.testing.assert_allclose(mx, g["enc_cls_max"], atol=3e-4)
    miss = 0
    for b in range(len(frames)):
        m, n, ws, wb = match_detections(g["labels"][b], g["boxes"][b], g["scores"][b], labels[b], boxes[b], scores[b], 1e-3, 1e-2)
        print(f"{name}[{b}] matched {m}/{n} worst dscore={ws:.2e} dbox={wb:.2e}px")
        miss += n - m
        assert m >= n - 1, (m, n)          # measured 300/300 on every frame; a selection flip at the K-th / K+1-th near-tie (gap ~1e-5) may move a row
    assert miss <= 1
    eng.close()


def x3_check(name, arch, w, input_size, eng, frames, ref_topk, ref_scores_all, refs, score_tol, cut_tol=5e-5):
    """f16x3 at the north-star tolerance (1e-3 on scores, 1e-2 px on boxes) with no slack on rows, at every input size.  The model has two
    top-k cuts, and a near-tie AT a cut may fall either way under any rounding pattern (the oracle's and HF's disagree there too), so:
      * encoder query selection: where the engine's selected token set differs from the reference's, every exchanged token must be within
        2 x the measured score tolerance of the reference's rank-Q score - and the frame is then compared, ALL rows at the full
        tolerance, with the oracle run on the engine's own selection (one exchanged query moves every row through the decoder's
        self-attention, so the reference's rows are not the yardstick for that frame; the oracle under the same selection is);
      * post-processor top-Q over the Q x C (query, class) scores: a reference row without a partner must sit within `cut_tol` (3 x the
        worst measured score error) of the reference's LOWEST kept score, i.e. at the cut;
      * everything else must match - free-running on frames whose token sets agree, and on every frame with the selection forced to the
        reference's.
    refs: list of (tag, labels, boxes, scores) per reference; ref_topk [B,Q]; ref_scores_all [B,S] = the reference's enc_cls_max."""
    Q = arch.num_queries

    def rows(tag_phase, b, use_refs, labels, boxes, scores):
        for tag, rl, rb, rs in use_refs:
            m, n, ws, wb, un = match_detections(rl[b], rb[b], rs[b], labels[b], boxes[b], scores[b], 1e-3, 1e-2, return_unmatched=True)
            cut = float(np.min(rs[b]))
            off_cut = [i for i in un if float(rs[b][i]) - cut > cut_tol]
            print(f"{name}[{b}] f16x3 {tag_phase} vs {tag}: matched {m}/{n} worst dscore={ws:.2e} dbox={wb:.2e}px; "
                  f"unmatched at the top-{Q} cut: {len(un) - len(off_cut)}, elsewhere: {len(off_cut)}")
            assert not off_cut and len(un) <= 1, (tag, b, m, n, [float(rs[b][i]) - cut for i in un])   # measured (round 5, 88 frame checks): 0 everywhere; one near-tie at the cut is the whole allowance

    labels, boxes, scores = eng.infer_raw(frames)
    mx = eng.debug_tensor("enc_cls_max")[:, :, 0, 0]
    tk = eng.debug_tensor("topk")[:, :, 0, 0].astype(np.int64)                 # the tokens the decoder ran on
    flips = []
    for b in range(len(frames)):
        mine = set(tk[b].tolist())
        assert len(mine) == Q
        ref = set(np.asarray(ref_topk[b]).tolist())
        diff = mine ^ ref
        kth = np.sort(ref_scores_all[b])[-Q]
        # exact ties with the rank-Q score (masked anchors share one score: whole groups of identical rows) are interchangeable, not flips
        flips.append(len([t for t in diff if float(ref_scores_all[b][t]) != float(kth)]) // 2)
        if diff:
            worst = max(abs(float(ref_scores_all[b][t]) - float(kth)) for t in diff)
            print(f"{name}[{b}] selection differs in {len(diff) // 2} token(s); farthest from the rank-{Q} score: {worst:.2e}")
            assert worst <= 2 * score_tol, (b, worst)
            # the oracle on THIS selection (frame b alone: frames are independent units)
            torch.set_num_threads(min(16, len(__import__('os').sched_getaffinity(0))))
            xb, sz = orc.preprocess(frames[b], input_size)
            with torch.no_grad():
                l2, b2, s2 = orc.model_forward(arch, w, xb, [sz], force_topk=tk[b:b + 1])
            own = [("oracle on the engine's selection", {b: l2[0].numpy()}, {b: b2[0].numpy()}, {b: s2[0].numpy()})]
            rows("free-running", b, own, labels, boxes, scores)
        else:
            rows("free-running", b, refs, labels, boxes, scores)
        assert (np.diff(scores[b]) <= 0).all(), "scores must be descending"
    assert sum(flips) <= 1, flips           # measured (round 5): none on any case - the only differing selections are exact ties of masked anchors, which are not flips
    eng.force_topk(np.asarray(ref_topk))
    labels, boxes, scores = eng.infer_raw(frames)
    eng.force_topk(None)
    for b in range(len(frames)):
        rows("reference selection", b, refs, labels, boxes, scores)


X3_CASES = ["t_tinyc_160x224", "c1_r18_640_bs1", "c1_r18_640_scene", "c1_r18_640_resize", "c2_r50_640_scene_bs2"]


@pytest.mark.parametrize("name", X3_CASES)
def test_f16x3_engine_matches_oracle_and_golden(name):
    """The default engine (precision f16x3: hi/lo fp16 pairs, three MFMAs per product) is held to the north-star tolerance with
    its own free-running query selection: 1e-3 on scores, 1e-2 px on boxes, against the oracle AND the HF fixtures."""
    arch, wseed, input_size, frames, g = load_case(name)
    w = weights_for(arch, wseed)
    (ol, ob, osc), col = oracle_run(arch, w, frames, input_size)
    eng = make_engine(arch, w, frames, input_size, "f16x3")
    labels, boxes, scores = eng.infer_raw(frames)
    x = nchw(eng.debug_tensor("input"))[:, :3]
    np.testing.assert_array_equal(x, col["input"].numpy())                 # the split engine keeps fp32 pixels: bit-exact preprocessing
    for i in range(3):
        e = rel_err(nchw(eng.debug_tensor(f"backbone{i}")), col[f"backbone{i}"].numpy())
        print(f"{name} backbone{i} rel l2 err {e:.2e}")
        assert e < 4e-5, (f"backbone{i}", e)
    for i in range(3):
        e = rel_err(nchw(eng.debug_tensor(f"enc{i}")), col[f"enc{i}"].numpy())
        print(f"{name} enc{i} rel l2 err {e:.2e}")
        assert e < 1e-4, (f"enc{i}", e)
    mx = eng.debug_tensor("enc_cls_max")[:, :, 0, 0]
    print(f"{name} enc score max abs err {np.abs(mx - col['enc_cls_max'].numpy()).max():.2e}")
    np.testing.assert_allclose(mx, col["enc_cls_max"].numpy(), atol=5e-4)
    refs = [("oracle", [t.numpy() for t in ol], [t.numpy() for t in ob], [t.numpy() for t in osc]), ("hf", g["labels"], g["boxes"], g["scores"])]
    x3_check(name, arch, w, input_size, eng, frames, col["topk"].numpy(), col["enc_cls_max"].numpy(), refs, 5e-4)
    eng.close()


@pytest.mark.parametrize("name", ["c2_r50_640_bs8", "c3_r101_1280_bs1", "c3_r101_1280_bs4", "c4_r18_1920_bs1"])
def test_f16x3_engine_full_size_configs_against_hf_fixtures(name):
    """BASELINE configs 2 (R50 640 bs8 = the benchmark frames) and 3 (R101 1280) on the default engine, hipGraph, against the
    committed HF outputs at the north-star tolerance; c4 = the reference's largest advertised input (config/config.yaml:122): a 1080p
    frame stretched to 1920 x 1920 through the PIL-exact resampler, R18, 75 600 memory tokens, 1e-2 px = 5.2e-6 of the frame."""
    arch, wseed, input_size, frames, g = load_case(name)
    w = weights_for(arch, wseed)
    eng = make_engine(arch, w, frames, input_size, "f16x3", use_graph=True)
    for _ in range(2):
        labels, boxes, scores = eng.infer_raw(frames)
    mx = eng.debug_tensor("enc_cls_max")[:, :, 0, 0]
    print(f"{name} enc score max abs err {np.abs(mx - g['enc_cls_max']).max():.2e}")
    np.testing.assert_allclose(mx, g["enc_cls_max"], atol=5e-4)
    # 1280-pixel frames: 1e-2 px is 7.8e-6 of the frame.  fp16 pairs carry 2^-22 (round 2's bf16 pairs, 2^-17 = 7.6e-6, left 0-4 rows of 300
    # between 1e-2 and 1.75e-2 px on R101 1280); no slack on rows at any size.
    x3_check(name, arch, w, input_size, eng, frames, g["topk"], g["enc_cls_max"], [("hf", g["labels"], g["boxes"], g["scores"])], 5e-4)
    eng.close()


def test_full_size_properties_bf16_bs8():
    """Size-independent properties at the benchmark configuration (R50 640 bs8 bf16, hipGraph):
    determinism across replays, batch == singles, frame-order equivariance, sorted scores, label range."""
    from telescope_cam_detection_amd.synth import noise_frame, scene_frame
    from telescope_cam_detection_amd.arch import ARCHS
    arch = ARCHS["r50"]
    w = weights_for(arch, 0)
    frames = [scene_frame(50 + i, 640, 640) if i % 2 else noise_frame(50 + i, 640, 640) for i in range(8)]
    eng = make_engine(arch, w, frames, (640, 640), "bf16", use_graph=True)
    a = eng.infer_raw(frames)
    b = eng.infer_raw(frames)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)                                  # graph replay is deterministic
    labels, boxes, scores = a
    assert labels.min() >= 0 and labels.max() < 80 and np.isfinite(boxes).all()
    assert (np.diff(scores, axis=1) <= 0).all() and (scores >= 0).all() and (scores <= 1).all()
    perm = [3, 0, 7, 1, 6, 2, 5, 4]
    pl, pb, ps = eng.infer_raw([frames[i] for i in perm])
    for j, i in enumerate(perm):                                             # frames are independent units
        m, n, ws, wb = match_detections(labels[i], boxes[i], scores[i], pl[j], pb[j], ps[j], 1e-6, 1e-4)
        assert m == n, (i, m, n, ws, wb)
    sl, sb, ss = eng.infer_raw([frames[2]])                                  # bs=1 plan (different tile dispatch)
    m, n, ws, wb = match_detections(labels[2], boxes[2], scores[2], sl[0], sb[0], ss[0], 2e-3, 0.5)
    print(f"bs8 vs bs1 (bf16): matched {m}/{n} worst dscore={ws:.2e} dbox={wb:.2e}px")
    assert m >= n - 30
    eng.close()


def test_full_size_properties_f16x3_bs8():
    """The default engine at the benchmark configuration (R50 640 bs8, hipGraph): replay determinism, frame-order equivariance (bit-exact:
    frames are independent units and every kernel choice depends on per-image extents or on the grid as a whole, never on a frame's
    slot), sorted scores, label range, and batch invariance: a frame of the bs-8 call equals the bs-1 call of the same handle bit for bit
    (the bs-1 plan picks other tile shapes on the small grids; every kernel keeps an output's K order)."""
    from telescope_cam_detection_amd.synth import noise_frame, scene_frame
    from telescope_cam_detection_amd.arch import ARCHS
    arch = ARCHS["r50"]
    w = weights_for(arch, 0)
    frames = [scene_frame(50 + i, 640, 640) if i % 2 else noise_frame(50 + i, 640, 640) for i in range(8)]
    eng = make_engine(arch, w, frames, (640, 640), "f16x3", use_graph=True)
    a = eng.infer_raw(frames)
    b = eng.infer_raw(frames)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    labels, boxes, scores = a
    assert labels.min() >= 0 and labels.max() < 80 and np.isfinite(boxes).all()
    assert (np.diff(scores, axis=1) <= 0).all() and (scores >= 0).all() and (scores <= 1).all()
    perm = [3, 0, 7, 1, 6, 2, 5, 4]
    pl, pb, ps = eng.infer_raw([frames[i] for i in perm])
    for j, i in enumerate(perm):
        np.testing.assert_array_equal(labels[i], pl[j])
        np.testing.assert_array_equal(boxes[i], pb[j])
        np.testing.assert_array_equal(scores[i], ps[j])
    worst = 0.0
    for i in (2, 5):
        sl, sb, ss = eng.infer_raw([frames[i]])
        m, n, ws, wb, un = match_detections(labels[i], boxes[i], scores[i], sl[0], sb[0], ss[0], 1e-3, 1e-2, return_unmatched=True)
        exact = np.array_equal(labels[i], sl[0]) and np.array_equal(boxes[i], sb[0]) and np.array_equal(scores[i], ss[0])
        print(f"bs8[{i}] vs bs1 (f16x3): matched {m}/{n} worst dscore={ws:.2e} dbox={wb:.2e}px bit-exact={exact}")
        assert exact, (i, m, n, ws, wb)       # measured: bit-exact - the small-grid tile shapes of the bs-1 plan keep every output's K order
        worst = max(worst, wb)
    eng.close()


@pytest.mark.parametrize("aname,size,bs", [("r50", 640, 14), ("r18", 320, 16), ("r50", 256, 9)])
def test_f16x3_batch_invariance_at_large_batches_and_small_inputs(aname, size, bs):
    """A frame's result does not depend on the batch it travels in - bit for bit - also above the batch where round 2's fixed split-K slab
    ran out (13 at R50 640) and on small inputs, where tile shapes and the direct-kernel choice once followed the batch: the slice count
    and every kernel family are chosen from per-image extents, the tile kernels keep an output's K order."""
    from telescope_cam_detection_amd.synth import noise_frame, scene_frame
    from telescope_cam_detection_amd.arch import ARCHS
    arch = ARCHS[aname]
    w = weights_for(arch, 0)
    frames = [scene_frame(70 + i, size, size) if i % 2 else noise_frame(70 + i, size, size) for i in range(bs)]
    eng = make_engine(arch, w, frames, (size, size), "f16x3", use_graph=True)
    labels, boxes, scores = eng.infer_raw(frames)
    for i in (0, bs // 2, bs - 1):
        sl, sb, ss = eng.infer_raw([frames[i]])
        assert np.array_equal(labels[i], sl[0]) and np.array_equal(boxes[i], sb[0]) and np.array_equal(scores[i], ss[0]), (aname, size, bs, i)
    half = eng.infer_raw(frames[: bs // 2])
    for x, y in zip((labels, boxes, scores), half):
        np.testing.assert_array_equal(x[: bs // 2], y)
    eng.close()


def test_side_stream_plan_equals_the_serial_plan():
    """side_stream (query selection beside the value projection, decoder input projections beside the PAN path: fork / join edges in the
    hipGraph) runs the same kernels on the same data as the one-stream plan: bit-identical outputs, eager and replayed."""
    from telescope_cam_detection_amd import _capi
    from telescope_cam_detection_amd.synth import noise_frame, scene_frame
    from telescope_cam_detection_amd.arch import ARCHS
    arch = ARCHS["r50"]
    w = weights_for(arch, 0)
    frames = [scene_frame(70, 640, 640), noise_frame(71, 640, 640), scene_frame(72, 640, 640)]
    outs = []
    for v in (0, 1, 3, 7):                        # bit 2: the encoder input projections beside stages 2 / 3 and AIFI
        _capi.debug_option("side_stream", v)
        e = make_engine(arch, w, frames, (640, 640), "f16x3", use_graph=True)
        for _ in range(3):
            o = e.infer_raw(frames)
        outs.append(o)
        e.close()
    _capi.debug_option("reset", 0)
    for o in outs[1:]:
        for x, y in zip(outs[0], o):
            np.testing.assert_array_equal(x, y)


def test_f16x3_fused_reduce_convs_match_separate_launches():
    """f16x3: the streaming expand convs of stage 0 carry the next block's reduce conv (also s0's last block -> stage 1's first c1):
    three launches fewer, stage outputs equal to the unfused plan to the engine's rounding (only that GEMM's accumulation order differs)."""
    from telescope_cam_detection_amd import _capi
    from telescope_cam_detection_amd.synth import noise_frame, scene_frame
    from telescope_cam_detection_amd.arch import ARCHS
    arch = ARCHS["r50"]
    w = weights_for(arch, 0)
    frames = [scene_frame(90, 640, 640), noise_frame(91, 640, 640)]
    outs, launches = [], []
    for v in (0, 1):
        _capi.debug_option("c1_fuse", v)
        e = make_engine(arch, w, frames, (640, 640), "f16x3")
        e.infer_raw(frames)
        outs.append([e.debug_tensor(f"backbone{i}").astype(np.float64).copy() for i in range(3)])
        launches.append(len(e.profile(2, 1)))
        e.close()
    _capi.debug_option("c1_fuse", 1)
    # three reduce convs fewer; and stage 0's last conv only carries the vd-shortcut average (avg_fuse) in its follower form: one avg-pool launch more without
    assert launches[0] - launches[1] == 4, launches
    for a, b in zip(*outs):
        rel = np.linalg.norm(a - b) / np.linalg.norm(a)
        print(f"f16x3 fused reduce convs: rel l2 {rel:.2e}")
        assert rel < 2e-5


def test_split_bf16_self_attention_matches_fp32_mfma_attention():
    """The bf16 engine's self-attention on hi/lo bf16 MFMAs (K / V stored as split fragments) against the exact fp32-MFMA attention
    of the same engine, each toggled ALONE so that everything upstream is bit-identical (downstream of a changed tensor the bf16 conv
    stack amplifies any 1e-6 difference to its own rounding noise, which would hide the comparison): AIFI output, then decoder logits /
    boxes with the encoder untouched.  3 images of 300 queries = an odd number (19) of key tiles per image."""
    from telescope_cam_detection_amd import _capi
    from telescope_cam_detection_amd.synth import noise_frame, scene_frame
    from telescope_cam_detection_amd.arch import ARCHS
    arch = ARCHS["r50"]
    w = weights_for(arch, 0)
    frames = [scene_frame(110 + i, 640, 640) if i % 2 else noise_frame(110 + i, 640, 640) for i in range(3)]
    got = {}
    for v in (0, 1, 2):
        _capi.debug_option("attn_split", v)
        e = make_engine(arch, w, frames, (640, 640), "bf16")
        raw = e.infer_raw(frames)
        got[v] = dict(aifi=e.debug_tensor("aifi_out").astype(np.float64).copy(), logits=e.debug_tensor("logits").astype(np.float64).copy(),
                      boxes=np.asarray(raw[1], np.float64), scores=np.asarray(raw[2], np.float64))
        e.close()
    _capi.debug_option("reset", 0)
    rel = lambda x, y: np.linalg.norm(x - y) / np.linalg.norm(x)
    e_aifi = rel(got[0]["aifi"], got[1]["aifi"])
    e_log = rel(got[0]["logits"], got[2]["logits"])
    d_sc = np.abs(got[0]["scores"] - got[2]["scores"]).max()
    # boxes by rank only where the neighbouring scores are distinct (equal scores may swap ranks)
    sc = got[0]["scores"]
    distinct = np.ones_like(sc, bool)
    distinct[:, 1:] &= np.abs(np.diff(sc, axis=1)) > 1e-4
    distinct[:, :-1] &= np.abs(np.diff(sc, axis=1)) > 1e-4
    d_box = np.abs(got[0]["boxes"] - got[2]["boxes"])[distinct].max()
    print(f"split-bf16 attention vs fp32 MFMA: AIFI out rel l2 {e_aifi:.2e}; decoder logits rel l2 {e_log:.2e}, max dbox {d_box:.2e}px, max dscore {d_sc:.2e}")
    # AIFI: |S| reaches tens there, and a 2^-16 relative product error is an absolute error of the softmax argument (1.7e-4 measured, 20x
    # below the bf16 conv stack's noise floor); the decoder's logits are small
    assert e_aifi < 5e-4 and e_log < 1e-4 and d_box < 5e-2 and d_sc < 1e-4


def test_throughput_profile_equals_latency_profile():
    """rtd_config.profile only changes which conv tile runs a layer (256- instead of 128-pixel tiles from 100 blocks on): the
    K order of every output is the same, so the detections must agree to the last bit of the accumulation."""
    from telescope_cam_detection_amd import _capi
    from telescope_cam_detection_amd.synth import noise_frame, scene_frame
    from telescope_cam_detection_amd.arch import ARCHS
    arch = ARCHS["r50"]
    w = weights_for(arch, 0)
    frames = [scene_frame(70 + i, 640, 640) if i % 2 else noise_frame(70 + i, 640, 640) for i in range(4)]
    e1 = make_engine(arch, w, frames, (640, 640), "bf16", use_graph=True, profile=_capi.PROFILE_LATENCY)
    e2 = make_engine(arch, w, frames, (640, 640), "bf16", use_graph=True, profile=_capi.PROFILE_THROUGHPUT)
    a = e1.infer_raw(frames)
    b = e2.infer_raw(frames)
    k1 = {p["name"]: p["ms"] for p in e1.profile(4, 2)}
    k2 = {p["name"]: p["ms"] for p in e2.profile(4, 2)}
    assert k1.keys() == k2.keys()
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    e1.close(); e2.close()


@pytest.mark.parametrize("name", ["t_tiny_160", "c1_r18_640_scene"])
def test_graph_replay_equals_eager(name):
    arch, wseed, input_size, frames, g = load_case(name)
    w = weights_for(arch, wseed)
    e1 = make_engine(arch, w, frames, input_size, "fp32", use_graph=False)
    e2 = make_engine(arch, w, frames, input_size, "fp32", use_graph=True)
    a = e1.infer_raw(frames)
    for _ in range(3):                      # capture, then two replays
        b = e2.infer_raw(frames)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    e1.close(); e2.close()


def test_fused_decoder_equals_per_op_decoder():
    """decoder.hip's one-kernel-per-layer path against the one-launch-per-op path (both exact fp32)."""
    from telescope_cam_detection_amd import _capi
    arch, wseed, input_size, frames, g = load_case("c1_r18_640_scene")
    w = weights_for(arch, wseed)
    outs = []
    for fused in (0, 1):
        _capi.debug_option("dec_fused", fused)
        eng = make_engine(arch, w, frames, input_size, "fp32")
        outs.append(eng.infer_raw(frames) + (eng.debug_tensor(f"dec{arch.dec_layers - 1}.hs"), eng.debug_tensor("ref")))
        eng.close()
    _capi.debug_option("dec_fused", 1)
    (l0, b0, s0, h0, r0), (l1, b1, s1, h1, r1) = outs
    np.testing.assert_allclose(h1, h0, atol=5e-5, rtol=1e-4)
    np.testing.assert_allclose(r1[..., :4], r0[..., :4], atol=2e-6)
    for b in range(len(frames)):
        m, n, ws, wb = match_detections(l0[b], b0[b], s0[b], l1[b], b1[b], s1[b], 1e-5, 2e-3)
        assert m == n, (m, n, ws, wb)


BF16_CASES = ["t_tiny_160", "t_tinyb_192x128", "c1_r18_640_scene", "c1_r18_640_resize", "c2_r50_640_scene_bs2"]


@pytest.mark.parametrize("name", BF16_CASES)
def test_bf16_engine_stagewise(name):
    arch, wseed, input_size, frames, g = load_case(name)
    w = weights_for(arch, wseed)
    (ol, ob, osc), col = oracle_run(arch, w, frames, input_size)
    eng = make_engine(arch, w, frames, input_size, "bf16")
    eng.infer_raw(frames)
    x = nchw(eng.debug_tensor("input"))[:, :3]
    assert np.abs(x - col["input"].numpy()).max() <= 2 ** -8          # one bf16 rounding of [0,1] values
    for i in range(3):
        e = rel_err(nchw(eng.debug_tensor(f"backbone{i}")), col[f"backbone{i}"].numpy())
        print(f"{name} backbone{i} rel l2 err {e:.2e}")
        assert e < 2e-2, (f"backbone{i}", e)
    for i in range(3):
        e = rel_err(nchw(eng.debug_tensor(f"enc{i}")), col[f"enc{i}"].numpy())
        print(f"{name} enc{i} rel l2 err {e:.2e}")
        assert e < 3e-2, (f"enc{i}", e)
    mx = eng.debug_tensor("enc_cls_max")[:, :, 0, 0]
    print(f"{name} enc score max abs err {np.abs(mx - col['enc_cls_max'].numpy()).max():.2e}")
    # decoder given the oracle's query selection: isolates bf16 feature noise from selection flips
    eng.force_topk(col["topk"].numpy())
    labels, boxes, scores = eng.infer_raw(frames)
    eng.force_topk(None)
    tot_m = tot_n = 0
    for b in range(len(frames)):
        m, n, ws, wb = match_detections(ol[b].numpy(), ob[b].numpy(), osc[b].numpy(), labels[b], boxes[b], scores[b], 2e-2, 2.0)
        print(f"{name}[{b}] bf16 (forced selection) matched {m}/{n} worst dscore={ws:.2e} dbox={wb:.2e}px")
        tot_m += m; tot_n += n
        assert n - m <= max(6, int(0.055 * n)), (b, m, n)      # measured: <= 11 of 300 (R18 / R50), <= 4 of 50 (tiny) miss at 2e-2 / 2 px; bound = 1.5x
    assert tot_m >= 0.95 * tot_n
    # free-running selection (what a user of precision="bf16" gets): overlap of the selected token set with the oracle's, and matched
    # rows at the same 2e-2 / 2 px.  bf16 score noise (3.5e-2 on the encoder logits) reorders near-ties around rank Q, so some queries
    # start from other tokens: the bounds are 1.5x the measured misses, and they are why bf16 is an opt-in, not the default engine.
    labels, boxes, scores = eng.infer_raw(frames)
    mx = eng.debug_tensor("enc_cls_max")[:, :, 0, 0]
    Q = arch.num_queries
    tot_m = tot_n = 0
    for b in range(len(frames)):
        mine = set(np.argsort(-mx[b], kind="stable")[:Q].tolist())
        ref = set(col["topk"][b].numpy().tolist())
        ov = len(mine & ref) / Q
        m, n, ws, wb = match_detections(ol[b].numpy(), ob[b].numpy(), osc[b].numpy(), labels[b], boxes[b], scores[b], 2e-2, 2.0)
        print(f"{name}[{b}] bf16 (free-running) token overlap {ov:.3f}, matched {m}/{n} worst dscore={ws:.2e} dbox={wb:.2e}px")
        tot_m += m; tot_n += n
        # measured: overlap >= 0.96; misses <= 16 of 300 (R18 / R50), <= 6 of 50 (tiny archs); bounds = 1.5x the measured misses
        assert ov >= 0.94, (b, ov)
        assert n - m <= max(9, int(0.08 * n)), (b, m, n)
    assert tot_m >= 0.90 * tot_n
    eng.close()


def test_bf16_engine_on_the_benchmark_frames_against_hf_fixture():
    """precision="bf16" on BASELINE config 2's own frames (R50 640 bs8, seeds 2000-2007), free-running, against the committed HF
    outputs: what the faster opt-in engine delivers, with the measured figures as bounds (the default f16x3 engine is held to
    1e-3 / 1e-2 px on the same fixture above)."""
    arch, wseed, input_size, frames, g = load_case("c2_r50_640_bs8")
    w = weights_for(arch, wseed)
    eng = make_engine(arch, w, frames, input_size, "bf16", use_graph=True)
    for _ in range(2):
        labels, boxes, scores = eng.infer_raw(frames)
    mx = eng.debug_tensor("enc_cls_max")[:, :, 0, 0]
    print(f"bf16 c2_r50_640_bs8 enc score max abs err {np.abs(mx - g['enc_cls_max']).max():.2e}")
    assert np.abs(mx - g["enc_cls_max"]).max() < 4.2e-2                      # measured 2.7e-2
    tot_m = tot_n = 0
    Q = arch.num_queries
    for b in range(len(frames)):
        ov = len(set(np.argsort(-mx[b], kind="stable")[:Q].tolist()) & set(g["topk"][b].tolist())) / Q
        m, n, ws, wb = match_detections(g["labels"][b], g["boxes"][b], g["scores"][b], labels[b], boxes[b], scores[b], 2e-2, 2.0)
        print(f"bf16 c2_r50_640_bs8[{b}] token overlap {ov:.3f}, matched {m}/{n} worst dscore={ws:.2e} dbox={wb:.2e}px")
        tot_m += m; tot_n += n
        assert ov >= 0.955 and n - m <= 29, (b, ov, m, n)                      # measured: overlap >= 0.970, <= 19 of 300 rows miss
    assert tot_m >= 0.92 * tot_n
    eng.close()


def test_bf16_engine_r101_1280_bs4_against_hf_fixture():
    """BASELINE config 3 at its full batch on the bf16 engine (the bs-4 plan takes other tiles than bs 1): stage-level sanity +
    matched rows at bf16 tolerances against the committed HF outputs."""
    arch, wseed, input_size, frames, g = load_case("c3_r101_1280_bs4")
    w = weights_for(arch, wseed)
    eng = make_engine(arch, w, frames, input_size, "bf16", use_graph=True)
    for _ in range(2):
        labels, boxes, scores = eng.infer_raw(frames)
    mx = eng.debug_tensor("enc_cls_max")[:, :, 0, 0]
    print(f"bf16 c3_r101_1280_bs4 enc score max abs err {np.abs(mx - g['enc_cls_max']).max():.2e}")
    assert np.isfinite(boxes).all() and (np.diff(scores, axis=1) <= 0).all()
    assert np.abs(mx - g["enc_cls_max"]).max() < 6e-2                        # measured 3.9e-2
    tot_m = tot_n = 0
    for b in range(len(frames)):
        m, n, ws, wb = match_detections(g["labels"][b], g["boxes"][b], g["scores"][b], labels[b], boxes[b], scores[b], 2e-2, 4.0)
        print(f"bf16 c3_r101_1280_bs4[{b}] matched {m}/{n} worst dscore={ws:.2e} dbox={wb:.2e}px")
        tot_m += m; tot_n += n
        assert n - m <= 32, (b, m, n)                                          # measured: <= 21 of 300 rows miss at 2e-2 / 4 px
    assert tot_m >= 0.90 * tot_n
    eng.close()


def test_detector_class_end_to_end():
    """The drop-in class: dict schema, ordering, threshold + wildlife filter, batch == singles."""
    from telescope_cam_detection_amd.rtdetr_detector import RTDETRDetector
    arch, wseed, input_size, frames, g = load_case("t_tinyb_192x128")
    w = weights_for(arch, wseed)
    det = RTDETRDetector(config_path="tinyb", model_path=f"synthetic:tinyb:{wseed}", device="cuda:0", conf_threshold=0.2,
                         input_size=input_size, wildlife_only=False, precision="fp32", max_batch=4)
    assert det.detect(frames[0]) == []          # not loaded -> [] (src/rtdetr_detector.py:248-250)
    assert det.detect_batch(frames) == [[], [], []]
    assert det.load_model() is True
    assert det.detect_batch([]) == []
    batch = det.detect_batch(frames)
    want = orc.detect_batch(arch, w, frames, input_size, 0.2, False)
    assert len(batch) == len(frames)
    for got, ref in zip(batch, want):
        assert abs(len(got) - len(ref)) <= 2
        for d in got:
            assert set(d) == {"class_id", "class_name", "confidence", "bbox"} and set(d["bbox"]) == {"x1", "y1", "x2", "y2", "area"}
            assert isinstance(d["bbox"]["area"], int) and d["confidence"] >= 0.2
        confs = [d["confidence"] for d in got]
        assert confs == sorted(confs, reverse=True)
        rl = np.array([d["class_id"] for d in ref]); rs = np.array([d["confidence"] for d in ref])
        rb = np.array([[d["bbox"][k] for k in ("x1", "y1", "x2", "y2")] for d in ref]).reshape(-1, 4)
        gl = np.array([d["class_id"] for d in got]); gs = np.array([d["confidence"] for d in got])
        gb = np.array([[d["bbox"][k] for k in ("x1", "y1", "x2", "y2")] for d in got]).reshape(-1, 4)
        m, n, ws, wb = match_detections(rl, rb, rs, gl, gb, gs, 1e-3, 1e-2)
        assert m >= n - 2, (m, n)
    single = det.detect(frames[1])
    assert [d["class_id"] for d in single] == [d["class_id"] for d in batch[1]]
    # wildlife filter + torch tensor input on the device (src/rtdetr_detector.py:217-219)
    det.wildlife_only = True
    only = det.detect(torch.from_numpy(frames[0]).cuda())
    assert all(d["class_id"] in (0, 14, 15, 16, 21) for d in only)
    assert det.is_wildlife_relevant(14) and not det.is_wildlife_relevant(2)
    assert det.get_class_category(0) == "person" and det.get_class_category(16) == "mammal" and det.get_class_category(5) == "other"
    with pytest.raises(RuntimeError):
        det.model.to("cpu")


@pytest.mark.parametrize("shapes", [[(160, 160), (120, 200)], [(90, 70), (160, 160), (50, 300)], [(400, 40), (160, 160)]])
def test_mixed_frame_sizes_go_through_the_resampler_like_the_oracle(shapes):
    """A batch with at least one odd-sized frame sends every frame through the PIL-exact resampler; its horizontal-pass
    intermediate must fit the tallest frame of the batch (regression: it was sized from the resized frames only, an
    identity-sized frame taller than them wrote past it - a GPU memory fault in __graft_entry__.smoke)."""
    from oracle import rtdetr_oracle as orc
    from telescope_cam_detection_amd import _capi
    from telescope_cam_detection_amd.arch import ARCHS
    from telescope_cam_detection_amd.synth import scene_frame
    from telescope_cam_detection_amd.weights import fold_weights, pack_blob, synth_weights

    arch = ARCHS["tiny"]
    w = synth_weights(arch, 1)
    frames = [scene_frame(20 + i, h, ww) for i, (h, ww) in enumerate(shapes)]
    eng = _capi.Engine(arch, pack_blob(fold_weights(arch, w)), 0, _capi.PREC_FP32, len(frames), (160, 160), True)
    for _ in range(2):
        labels, boxes, scores = eng.infer_raw(frames)
    got = eng.debug_tensor("input")[:, :, :, :3]
    for i, f in enumerate(frames):
        x, _ = orc.preprocess(f, (160, 160))
        np.testing.assert_array_equal(got[i], x[0].permute(1, 2, 0).numpy())          # bit-exact preprocessing
    xs, sizes = zip(*[orc.preprocess(f, (160, 160)) for f in frames])
    ol, ob, osc = orc.model_forward(arch, w, torch.cat(xs, 0), list(sizes))
    for b in range(len(frames)):
        assert np.abs(np.sort(scores[b])[::-1][:10] - np.sort(osc[b].numpy())[::-1][:10]).max() < 1e-3
    eng.close()


def test_f16x3_fused_uint8_stem_matches_the_generic_stem():
    """f16x3: backbone.stem.0 straight from the uint8 frames (default, hi/lo pairs made on the fly) against the generic form (fp32
    NHWC-8 image + exact fp32-MFMA conv): same detections at the north-star tolerance; the first stage output agrees to split rounding."""
    from telescope_cam_detection_amd import _capi
    from telescope_cam_detection_amd.arch import ARCHS
    from telescope_cam_detection_amd.synth import scene_frame
    from telescope_cam_detection_amd.weights import fold_weights, pack_blob, synth_weights

    arch = ARCHS["r18"]
    w = synth_weights(arch, 3)
    blob = pack_blob(fold_weights(arch, w))
    frames = [scene_frame(70, 640, 640), scene_frame(71, 480, 600), scene_frame(72, 640, 640)]
    out, stem = {}, {}
    for fused in (0, 1):
        _capi.debug_option("stem_fused_split", fused)
        eng = _capi.Engine(arch, blob, 0, _capi.PREC_F16X3, 3, (640, 640), True)
        for _ in range(2):
            out[fused] = eng.infer_raw(frames)
        stem[fused] = eng.debug_tensor("stem").astype(np.float64)
        out[("input", fused)] = eng.debug_tensor("input")[:, :, :, :3]
        eng.close()
    np.testing.assert_array_equal(out[("input", 0)], out[("input", 1)])
    e = np.linalg.norm(stem[0] - stem[1]) / np.linalg.norm(stem[0])
    print(f"f16x3 fused stem vs generic: stem rel l2 {e:.2e}")
    assert e < 1e-5
    for b in range(len(frames)):
        m, n, ws, wb = match_detections(out[0][0][b], out[0][1][b], out[0][2][b], out[1][0][b], out[1][1][b], out[1][2][b], 1e-3, 1e-2)
        print(f"f16x3 fused stem vs generic [{b}]: matched {m}/{n} worst dscore={ws:.2e} dbox={wb:.2e}px")
        assert m >= n - 2, (b, m, n, ws, wb)


@pytest.mark.parametrize("aname,size,bs", [("tinyc", (160, 224), 3), ("r18", (352, 608), 2), ("r50", (640, 640), 2), ("r18", (1280, 1280), 1)])
def test_f16x3_fused_stem2_maxpool_equals_conv_then_pool(aname, size, bs):
    """stem.2 and the 3x3 / stride-2 max-pool in one pass (rows exchanged between the waves of a tile through LDS, the first pooled row and
    column of every tile completed from side buffers by k_pool_fixup) against conv -> pool as two launches: the pooled map and the final
    outputs bit for bit - max is exact and ReLU outputs are >= 0.  Ragged tiles in both directions (352 x 608 -> a 176 x 304 conv map:
    22 tile rows, 9.5 tile columns), a single tile row per wave pair, the benchmark size and 1280 px."""
    from telescope_cam_detection_amd import _capi
    from telescope_cam_detection_amd.arch import ARCHS
    from telescope_cam_detection_amd.synth import noise_frame, scene_frame
    from telescope_cam_detection_amd.weights import fold_weights, pack_blob, synth_weights

    arch = ARCHS[aname]
    blob = pack_blob(fold_weights(arch, synth_weights(arch, 5)))
    frames = [noise_frame(80 + i, size[0], size[1]) if i % 2 else scene_frame(80 + i, size[0], size[1]) for i in range(bs)]
    out, stem = {}, {}
    for fused in (0, 1):
        _capi.debug_option("stem_pool_fuse", fused)
        eng = _capi.Engine(arch, blob, 0, _capi.PREC_F16X3, bs, size, True)
        for _ in range(2):
            out[fused] = eng.infer_raw(frames)
        stem[fused] = eng.debug_tensor("stem")
        one = eng.infer_raw(frames[:1])                       # the bs-1 plan of the same handle
        for x, y in zip(out[fused], one):
            np.testing.assert_array_equal(x[:1], y)
        eng.close()
    _capi.debug_option("reset", 0)
    assert stem[0].shape == stem[1].shape and np.isfinite(stem[1]).all()
    np.testing.assert_array_equal(stem[0], stem[1])
    for x, y in zip(out[0], out[1]):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("aname,size,bs", [("r50", (640, 640), 2), ("r50", (512, 768), 3), ("r50", (352, 608), 1), ("r101", (1280, 1280), 1)])
def test_f16x3_fused_vd_shortcut_average_equals_the_avgpool_launch(aname, size, bs):
    """A stage's last expand conv also writes the 2 x 2 / stride-2 average of its output (2 x 16 patch tiles in the streaming kernel,
    ConvArgs::avg_y) = the next stage's vd-shortcut input, against the separate avg-pool launch: backbone maps and final outputs bit for
    bit (the same ((a + b) + (c + d)) * 0.25 on the represented values).  352 x 608 has an 88 x 152 stage-0 map (152 % 16 != 0): the fusion
    declines and both plans are the same."""
    from telescope_cam_detection_amd import _capi
    from telescope_cam_detection_amd.arch import ARCHS
    from telescope_cam_detection_amd.synth import noise_frame, scene_frame
    from telescope_cam_detection_amd.weights import fold_weights, pack_blob, synth_weights

    arch = ARCHS[aname]
    blob = pack_blob(fold_weights(arch, synth_weights(arch, 6)))
    frames = [noise_frame(90 + i, size[0], size[1]) if i % 2 else scene_frame(90 + i, size[0], size[1]) for i in range(bs)]
    out, maps = {}, {}
    for fused in (0, 1):
        _capi.debug_option("avg_fuse", fused)
        eng = _capi.Engine(arch, blob, 0, _capi.PREC_F16X3, bs, size, True)
        for _ in range(2):
            out[fused] = eng.infer_raw(frames)
        maps[fused] = [eng.debug_tensor(f"backbone{i}") for i in range(3)]
        one = eng.infer_raw(frames[:1])
        for x, y in zip(out[fused], one):
            np.testing.assert_array_equal(x[:1], y)
        eng.close()
    _capi.debug_option("reset", 0)
    for a, b in zip(maps[0], maps[1]):
        assert np.isfinite(b).all()
        np.testing.assert_array_equal(a, b)
    for x, y in zip(out[0], out[1]):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("aname,size,bs", [("r50", (640, 640), 3), ("r101", (320, 512), 2)])
def test_f16x3_unwritten_stage0_output_changes_nothing(aname, size, bs):
    """ConvArgs::y_dead (round 5): at the stage-0 / stage-1 boundary of a bottleneck net the stage-0 output's only readers - stage 1's first
    reduce conv and the vd-shortcut average - are fused into the launch that produces it, so its stores are dropped (210 MB per R50 bs-8 step).
    Against the plan that writes it: one launch list, backbone maps and final rows bit for bit, graph replay included."""
    from telescope_cam_detection_amd import _capi
    from telescope_cam_detection_amd.arch import ARCHS
    from telescope_cam_detection_amd.synth import noise_frame, scene_frame
    from telescope_cam_detection_amd.weights import fold_weights, pack_blob, synth_weights

    arch = ARCHS[aname]
    blob = pack_blob(fold_weights(arch, synth_weights(arch, 6)))
    frames = [noise_frame(60 + i, size[0], size[1]) if i % 2 else scene_frame(60 + i, size[0], size[1]) for i in range(bs)]
    out, maps, nbytes, nops = {}, {}, {}, {}
    for dead in (0, 1):
        _capi.debug_option("dead_out", dead)
        eng = _capi.Engine(arch, blob, 0, _capi.PREC_F16X3, bs, size, True)
        for _ in range(3):
            out[dead] = eng.infer_raw(frames)
        maps[dead] = [eng.debug_tensor(f"backbone{i}") for i in range(3)]
        prof = eng.profile(bs, 1)
        nops[dead] = len(prof)
        nbytes[dead] = sum(p["bytes"] for p in prof if p["name"] == f"backbone.s0.b{arch.depths[0] - 1}.c3")
        eng.close()
    _capi.debug_option("reset", 0)
    assert nops[0] == nops[1]
    h4, w4 = size[0] // 4, size[1] // 4
    assert nbytes[0] - nbytes[1] == bs * h4 * w4 * 256 * 4                  # exactly the stage-0 output, 4 bytes per channel
    for a, b in zip(maps[0], maps[1]):
        np.testing.assert_array_equal(a, b)
    for x, y in zip(out[0], out[1]):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("aname,prec,size,bs", [("r50", "f16x3", (640, 640), 3), ("r18", "fp32", (320, 416), 2), ("tinyc", "bf16", (160, 224), 2)])
def test_fused_post_processor_equals_sigmoid_topk_gather(aname, prec, size, bs):
    """The post-processor (HF:image_processing_rt_detr.py:510-533) as ONE launch - sigmoid while the keys are loaded, top-k, the winners written
    as finished [label, score, x1, y1, x2, y2] rows (ops.hip TopkPost, round 5) - against the three-launch form: raw rows and the filtered
    `rtd_infer` rows bit for bit, graph replay included, frames of other sizes than the network's (the (w, h) scale differs per frame)."""
    from telescope_cam_detection_amd import _capi
    from telescope_cam_detection_amd.arch import ARCHS
    from telescope_cam_detection_amd.synth import noise_frame, scene_frame
    arch = ARCHS[aname]
    w = weights_for(arch, 2)
    frames = [scene_frame(30 + i, size[0] + 40 * i, size[1] - 16 * i) if i % 2 == 0 else noise_frame(30 + i, size[0], size[1]) for i in range(bs)]
    got = {}
    for fused in (0, 1):
        _capi.debug_option("post_fused", fused)
        eng = make_engine(arch, w, frames, size, prec, use_graph=True)
        for _ in range(3):
            raw = eng.infer_raw(frames)
        rows = eng.infer(frames, 0.0, False)
        names = [p["name"] for p in eng.profile(bs, 1)]
        assert ("post.fused" in names) == bool(fused) and ("post.topk" in names) != bool(fused), names
        got[fused] = (raw, rows)
        eng.close()
    _capi.debug_option("reset", 0)
    for x, y in zip(got[0][0], got[1][0]):
        np.testing.assert_array_equal(x, y)
    for x, y in zip(got[0][1], got[1][1]):
        assert len(x) == arch.num_queries and (x == y).all()


def test_non_square_input_with_partial_tiles_bf16_and_fp32():
    """416 x 736 (multiples of 32, but 208 x 368 and 104 x 184 are not multiples of the 8 x 32 / 128-pixel tiles): every conv
    kernel family meets ragged tiles.  fp32 engine vs oracle at the north-star tolerance, bf16 engine vs fp32 engine to bf16 noise."""
    from oracle import rtdetr_oracle as orc
    from telescope_cam_detection_amd import _capi
    from telescope_cam_detection_amd.arch import ARCHS
    from telescope_cam_detection_amd.synth import scene_frame
    from telescope_cam_detection_amd.weights import fold_weights, pack_blob, synth_weights

    torch.set_num_threads(8)
    arch = ARCHS["r18"]
    w = synth_weights(arch, 4)
    blob = pack_blob(fold_weights(arch, w))
    size = (416, 736)
    frames = [scene_frame(80 + i, size[0], size[1]) for i in range(3)]
    xs, sizes = zip(*[orc.preprocess(f, size) for f in frames])
    ol, ob, osc = orc.model_forward(arch, w, torch.cat(xs, 0), list(sizes))
    res = {}
    for prec in (_capi.PREC_FP32, _capi.PREC_BF16):
        eng = _capi.Engine(arch, blob, 0, prec, 3, size, True)
        for _ in range(2):
            res[prec] = eng.infer_raw(frames)
        eng.close()
    for b in range(3):
        l, bx, sc = (t[b] for t in res[_capi.PREC_FP32])
        m, n, ws, wb = match_detections(ol[b].numpy(), ob[b].numpy(), osc[b].numpy(), l, bx, sc, 1e-3, 1e-2)
        assert m == n, ("fp32", b, m, n, ws, wb)
        l2, bx2, sc2 = (t[b] for t in res[_capi.PREC_BF16])
        assert np.isfinite(bx2).all() and (np.diff(sc2) <= 0).all()
        m, n, ws, wb = match_detections(l, bx, sc, l2, bx2, sc2, 3e-2, 4.0)
        print(f"non-square bf16 vs fp32 [{b}]: matched {m}/{n} worst dscore={ws:.2e} dbox={wb:.2e}px")
        assert m >= n - 18, ("bf16", b, m, n, ws, wb)                            # measured: 10-12 of 300 miss at 3e-2 / 4 px


def test_f16x3_non_square_frames_with_ragged_tiles_against_the_oracle():
    """R50 at 480 x 608 (multiples of 32; 120 x 152, 60 x 76, 30 x 38 and 15 x 19 maps are not multiples of the 8 x 16 / 32- / 128-pixel tiles): the
    default engine's direct, streaming (fused follower, tiles straddling images), tiled, flexible and split-K kernels all meet ragged tiles inside
    the network.  Against the oracle at the north-star tolerance, free-running and with the oracle's query selection."""
    from telescope_cam_detection_amd.arch import ARCHS
    from telescope_cam_detection_amd.synth import noise_frame, scene_frame
    arch = ARCHS["r50"]
    w = weights_for(arch, 3)
    size = (480, 608)
    frames = [scene_frame(300, size[0], size[1]), noise_frame(301, size[0], size[1]), scene_frame(302, 360, 500)]      # the third one is resampled
    (ol, ob, osc), col = oracle_run(arch, w, frames, size)
    eng = make_engine(arch, w, frames, size, "f16x3", use_graph=True)
    eng.infer_raw(frames)
    for i in range(3):
        e = rel_err(nchw(eng.debug_tensor(f"enc{i}")), col[f"enc{i}"].numpy())
        print(f"480x608 enc{i} rel l2 err {e:.2e}")
        assert e < 1e-4
    refs = [("oracle", [t.numpy() for t in ol], [t.numpy() for t in ob], [t.numpy() for t in osc])]
    x3_check("r50_480x608", arch, w, size, eng, frames, col["topk"].numpy(), col["enc_cls_max"].numpy(), refs, 5e-4)
    eng.close()


@pytest.mark.parametrize("aname,size,bs", [("r34", (640, 640), 2), ("r101", (640, 640), 1), ("r18", (320, 320), 5), ("r50", (1280, 1280), 1), ("r50", (736, 1280), 2)])
def test_f16x3_other_backbones_and_sizes_against_the_oracle(aname, size, bs):
    """The variants the reference's config files name besides R50 (r18vd / r34vd basic blocks, r101vd), a small input / odd batch, R50 at 1280 px
    and on a 16:9 map (736 x 1280: every level non-square): default engine vs the oracle at the north-star tolerance (no golden file: the
    oracle itself is pinned to HF on the committed cases)."""
    from telescope_cam_detection_amd.arch import ARCHS
    from telescope_cam_detection_amd.synth import noise_frame, scene_frame
    arch = ARCHS[aname]
    w = weights_for(arch, 5)
    frames = [scene_frame(400 + i, size[0], size[1]) if i % 2 == 0 else noise_frame(400 + i, size[0], size[1]) for i in range(bs)]
    (ol, ob, osc), col = oracle_run(arch, w, frames, size)
    eng = make_engine(arch, w, frames, size, "f16x3", use_graph=True)
    refs = [("oracle", [t.numpy() for t in ol], [t.numpy() for t in ob], [t.numpy() for t in osc])]
    x3_check(f"{aname}_{size[0]}_bs{bs}", arch, w, size, eng, frames, col["topk"].numpy(), col["enc_cls_max"].numpy(), refs, 5e-4)
    eng.close()


def test_detector_keeps_working_after_the_callers_degrade_writes():
    """src/inference_engine_yolox.py:726-748 on the real engine: `detector.input_size = ...`, `detector.device = "cpu"` and
    `detector.model.to("cpu")` (raises, caught by the caller) - then detect, detect_batch and the pipelined pair must give what
    they gave before (the engine stays on its GPU)."""
    from telescope_cam_detection_amd.rtdetr_detector import RTDETRDetector
    arch, wseed, input_size, frames, g = load_case("t_tinyb_192x128")
    det = RTDETRDetector(config_path="tinyb", model_path=f"synthetic:tinyb:{wseed}", device="cuda:0", conf_threshold=0.1,
                         input_size=input_size, wildlife_only=False, precision="fp32", max_batch=4)
    assert det.load_model() is True
    before = det.detect_batch(frames)
    det.input_size = (96, 64)
    det.device = "cpu"
    with pytest.raises(RuntimeError):
        det.model.to("cpu")
    assert det.detect(frames[0]) == before[0]
    assert det.detect_batch(frames) == before
    ticket = det.detect_batch_async(frames)
    assert det.detect_batch_collect(ticket) == before
    x, wh = det.preprocess(frames[1])
    assert x.is_cuda and tuple(x.shape) == (1, 3) + tuple(input_size) and wh.tolist() == [[frames[1].shape[1], frames[1].shape[0]]]


def test_preprocess_returns_the_reference_tensor_bit_for_bit():
    """`RTDETRDetector.preprocess` (src/rtdetr_detector.py:206-236: BGR -> RGB, ToPILImage -> Resize -> ToTensor, orig_size [[w, h]]) as one
    small device launch (rtd_preprocess, round 5: no forward pass, no host round trip): equal to the oracle's PIL pipeline bit for bit for a
    frame of the network's size, a 720p and a 1080p frame, from host memory and from a device tensor - on every engine precision."""
    from oracle import rtdetr_oracle as orc
    from telescope_cam_detection_amd.rtdetr_detector import RTDETRDetector
    from telescope_cam_detection_amd.synth import noise_frame, scene_frame
    size = (320, 416)
    frames = [scene_frame(5, 320, 416), noise_frame(6, 720, 1280), scene_frame(7, 1080, 1920), noise_frame(8, 97, 131)]
    for prec in ("f16x3", "bf16"):
        det = RTDETRDetector(config_path="tinyc", model_path="synthetic:tinyc:1", device="cuda:0", input_size=size, precision=prec, max_batch=2)
        assert det.load_model(max_retries=1) is True
        for f in frames:
            want, (w_, h_) = orc.preprocess(f, size)
            for src in (f, torch.from_numpy(f).cuda(), torch.from_numpy(f)):
                x, wh = det.preprocess(src)
                assert x.is_cuda and x.dtype == torch.float32 and tuple(x.shape) == (1, 3) + size
                assert torch.equal(x.cpu(), want), (prec, f.shape)
                assert wh.tolist() == [[w_, h_]] == [[f.shape[1], f.shape[0]]]
        before = det.detect(frames[1])
        det.preprocess(frames[2])                                   # shares the engine's staging buffers: the next detect is unaffected
        assert det.detect(frames[1]) == before
        det.model.engine.close()


def test_out_of_memory_surfaces_as_torch_cuda_out_of_memory_error():
    """A batch whose activation arena cannot be allocated (R18 at 4096 x 4096, 64 frames: several hundred GB) makes the library
    return RTD_E_OOM, which the shim re-raises as torch.cuda.OutOfMemoryError out of detect_batch - the one exception the caller's
    recovery path handles (src/inference_engine_yolox.py:607-623).  Smaller batches keep working on the same handle afterwards."""
    from telescope_cam_detection_amd.rtdetr_detector import RTDETRDetector
    from telescope_cam_detection_amd.synth import scene_frame
    det = RTDETRDetector(config_path="r18", model_path="synthetic:r18:0", device="cuda:0", conf_threshold=0.0, input_size=(4096, 4096),
                         wildlife_only=False, precision="fp32", max_batch=64)
    assert det.load_model(max_retries=1) is True
    one = det.model.engine.arena_bytes()
    free, total = torch.cuda.mem_get_info(0)
    if one * 64 < 1.2 * total:
        pytest.skip(f"bs-64 arena ({one * 64 / 1e9:.0f} GB) would fit this GPU ({total / 1e9:.0f} GB)")
    small = [scene_frame(300 + i, 48, 64) for i in range(64)]
    with pytest.raises(torch.cuda.OutOfMemoryError):
        det.detect_batch(small)
    got = det.detect(small[0])                                # the bs-1 plan of the same handle still runs
    assert isinstance(got, list) and len(got) > 0


def test_device_frames_are_ordered_after_the_stream_that_produced_them():
    """a device-resident frame that torch has just computed on its current stream (here: a large flip + slice made
    contiguous inside the detector) must be complete before the engine's own non-blocking stream reads it."""
    from telescope_cam_detection_amd.rtdetr_detector import RTDETRDetector
    from telescope_cam_detection_amd.synth import scene_frame
    det = RTDETRDetector(config_path="tinyb", model_path="synthetic:tinyb:2", device="cuda:0", conf_threshold=0.05, input_size=(192, 128),
                         wildlife_only=False, precision="fp32", max_batch=2)
    assert det.load_model() is True
    big = scene_frame(400, 1500, 2000)
    want = det.detect(np.ascontiguousarray(big[::-1, 100:1900]))
    assert len(want) > 0
    dev = torch.from_numpy(big).cuda()
    torch.cuda.synchronize()
    for _ in range(5):
        filler = [torch.empty(64 << 20, device="cuda").normal_() for _ in range(4)]   # keeps torch's stream busy ahead of the producer ops
        view = dev.flip(0)[:, 100:1900]                                             # flip kernel + non-contiguous slice -> .contiguous() on torch's stream
        assert det.detect(view) == want
        t = det.detect_batch_async([view])
        assert det.detect_batch_collect(t) == [want]
        del filler


def test_no_captured_event_state_leaks_to_later_events_of_the_process():
    """Round 4: a plan's hipGraph is BUILT node by node (csrc/common.h GraphBuild) - the library never captures a stream, so neither a
    stream nor an event of the process can carry capture state because of it.  After engines with forked plans were graphed, replayed and
    closed, the handle reports no capture state and events created by anyone else (here: torch) record and query cleanly."""
    from telescope_cam_detection_amd import _capi
    from telescope_cam_detection_amd.arch import ARCHS
    from telescope_cam_detection_amd.synth import scene_frame
    arch = ARCHS["tinyc"]
    w = weights_for(arch, 3)
    frames = [scene_frame(40, 160, 224), scene_frame(41, 160, 224)]
    _capi.debug_option("side_stream", 7)
    for _ in range(3):
        eng = make_engine(arch, w, frames, (160, 224), "f16x3", use_graph=True)
        for n in (2, 1, 2):                                    # two plans graphed, then a replay
            eng.infer_raw(frames[:n])
        st = eng.stats()
        assert st["graphs"] == 2 and st["graph_launches"] == 3 and st["stream_capture_status"] == 0 and st["failed_calls"] == 0, st
        assert st["graph_nodes"] > 40                          # both lanes' kernels are nodes of the built graphs
        eng.close()
        evs = [torch.cuda.Event(enable_timing=(i % 2 == 0)) for i in range(64)]
        for ev in evs:
            ev.record()
        torch.cuda.synchronize()
        assert all(ev.query() for ev in evs)
        assert evs[0].elapsed_time(evs[2]) >= 0.0
