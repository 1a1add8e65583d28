"""GPU: every HIP kernel, called through the C ABI's kernel-level entry points, against a plain
PyTorch fp32 reference of the same op computed on the CPU (integer / index kernels: bit-exact)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {"bf16": (0, torch.bfloat16), "f32": (1, torch.float32)}


@pytest.fixture(scope="module")
def L():
    from telescope_cam_detection_amd import _capi
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return _capi.lib()


def ck(L, rc):
    assert rc == 0, (rc, L.rtd_last_error(None))


def nhwc(x, dt):
    return x.permute(0, 2, 3, 1).contiguous().to(dt).cuda()


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad, act, res_mode
    (2, 20, 20, 64, 64, 1, 1, 0, "relu", 0),
    (1, 33, 29, 32, 96, 3, 1, 1, "silu", 0),       # ragged M, Cout not a tile multiple
    (2, 40, 40, 64, 128, 3, 2, 1, "relu", 1),      # stride 2, pre-activation residual
    (1, 16, 16, 128, 256, 1, 1, 0, "none", 2),     # post-activation residual
    (2, 32, 32, 8, 32, 3, 2, 1, "relu", 0),        # stem-like: Cin 8 (small-C path), K=72 padded to 96
    (1, 24, 24, 16, 32, 3, 1, 1, "relu", 0),       # Cin 16 small-C path
    (2, 16, 16, 64, 256, 2, 2, 0, "none", 0),      # the folded AvgPool+1x1 shortcut (2x2 stride 2)
    (1, 300, 1, 256, 80, 1, 1, 0, "none", 0),      # token GEMM N=80
    (1, 300, 1, 256, 4, 1, 1, 0, "none", 0),       # token GEMM N=4
    (1, 300, 1, 8, 512, 1, 1, 0, "relu", 0),       # K=8 (query_pos layer 0)
    (1, 64, 64, 256, 256, 3, 1, 1, "gelu", 0),     # big enough for the 128x128 tile path
    (4, 80, 80, 128, 128, 3, 1, 1, "silu", 1),     # 128x128 tile, many blocks
    # shapes large enough for the v2 (128-pixel, LDS-staged epilogue) path
    (2, 120, 120, 64, 64, 1, 1, 0, "relu", 1),     # v2 BN=64, pre-activation residual
    (4, 100, 97, 128, 192, 3, 1, 1, "silu", 2),    # v2/v3 BN=128, ragged M, partial last N tile, post residual
    (2, 160, 160, 64, 256, 1, 1, 0, "relu", 1),    # stage-0 c3 shape
    (4, 208, 128, 256, 384, 3, 2, 1, "silu", 0),   # v2/v3 stride 2
    (1, 26000, 1, 256, 288, 1, 1, 0, "none", 0),   # v2 token GEMM, N = 288
    (2, 96, 96, 512, 64, 1, 1, 0, "gelu", 0),      # v2 BN=64, K = 512
    (1, 67200, 1, 256, 80, 1, 1, 0, "none", 0),    # enc_score_head shape: fp32 takes the LDS-DMA kernel with a partial N tile
    # narrow 3x3 layers on wide grids: the direct (filter-in-registers) kernel in bf16; ragged tiles in both directions
    (4, 60, 264, 32, 32, 3, 1, 1, "relu", 0),      # stem.1 shape family
    (4, 60, 264, 32, 64, 3, 1, 1, "silu", 0),      # stem.2
    (4, 68, 232, 64, 64, 3, 1, 1, "relu", 0),      # stage-0 c2 (two 32-channel groups)
    (2, 160, 160, 64, 64, 3, 1, 1, "none", 0),     # exact tiles
    (3, 100, 75, 64, 64, 3, 1, 1, "relu", 1),      # basic-block c2: pre-activation residual through the store slab, ragged tiles
    (2, 96, 128, 64, 64, 3, 1, 1, "silu", 2),      # post-activation residual
    # short K, many channel tiles, no residual: the A-stationary kernel (auto in mode 0 from 4 channel tiles on)
    (2, 64, 66, 128, 640, 1, 1, 0, "silu", 0),     # K = 128 (2 chunks), 5 channel tiles, ragged M
    (1, 8400, 1, 256, 1536, 1, 1, 0, "none", 0),   # one image of the value projection
    (3, 40, 40, 64, 520, 1, 1, 0, "relu", 0),      # K = 64 (1 chunk), partial last channel tile
    # thin 1x1 layers (K = 64 / 128, Cout a multiple of 64): the streaming filter-in-registers kernel (mode 9 forces it on small grids)
    (2, 50, 37, 128, 512, 1, 1, 0, "relu", 1),     # stage-1 c3 family: 8 channel groups, ragged last pixel tile
    (1, 45, 31, 64, 64, 1, 1, 0, "silu", 2),       # one channel group, four pixel sub-tiles per block, post-activation residual
    (3, 33, 33, 128, 128, 1, 1, 0, "none", 0),     # two channel groups
    (2, 96, 100, 64, 256, 1, 1, 0, "none", 0),     # stage-0 shortcut family (linear)
    # reducing 1x1 layers (K = 256 -> 64 / 128): streaming kernel with the filter in LDS
    (1, 70, 50, 256, 64, 1, 1, 0, "relu", 0),      # ragged last pixel tile
    (2, 40, 40, 256, 128, 1, 1, 0, "silu", 1),     # two channel groups per wave, residual through the slab
    (4, 128, 128, 256, 64, 1, 1, 0, "relu", 0),    # 2048 tiles: taken in mode 0 too
    (2, 40, 41, 256, 1024, 1, 1, 0, "relu", 1),    # stage-2 c3: eight 128-channel blocks in gridDim.y, residual, ragged tile
    (1, 30, 30, 256, 384, 1, 1, 0, "none", 0),     # three channel blocks
]


DUAL_CASES = [
    # B, H, W, Cin, C2, Cout, k, act: y = act(conv_kxk(x) + conv_1x1(x2) + bias), x2 at output resolution
    (2, 40, 36, 64, 64, 256, 1, "relu"),       # stage-0 block 0 of the bottleneck nets (64 + 64 channels: streaming kernel in bf16)
    (2, 30, 30, 128, 256, 512, 1, "relu"),     # stage-1 block 0: K = 384
    (1, 13, 11, 256, 512, 1024, 1, "relu"),    # stage-2 block 0, ragged pixel tile
    (2, 24, 20, 64, 64, 128, 3, "relu"),       # basic-block nets: 3x3 main conv (K1 = 576) + 1x1 shortcut, Cout one tile
    (1, 20, 20, 128, 64, 192, 3, "none"),      # partial last channel tile
    (4, 128, 128, 32, 64, 128, 1, "relu"),     # 512 tiles: large enough for the fp32 kernel (fp32 K-steps are 32 wide)
]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("case", DUAL_CASES)
def test_conv_dual(L, dt, case):
    """ConvArgs::x2 (the projection shortcut folded into a block's last conv) against conv(x) + conv1x1(x2) in fp64."""
    B, H, W, Cin, C2, Cout, k, act = case
    if dt == "bf16" and (Cin % 64 or C2 % 64):
        pytest.skip("bf16 K-steps are 64 channels wide")
    code, tdt = DT[dt]
    g = torch.Generator().manual_seed(700 + DUAL_CASES.index(case))
    x = torch.randn(B, Cin, H, W, generator=g)
    x2 = torch.randn(B, C2, H, W, generator=g)
    w1 = torch.randn(Cout, Cin, k, k, generator=g) * (1.0 / (Cin * k * k)) ** 0.5
    w2 = torch.randn(Cout, C2, 1, 1, generator=g) * (1.0 / C2) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    q = lambda t: t.to(tdt).double()
    y = F.conv2d(q(x), q(w1), b.double(), padding=k // 2) + F.conv2d(q(x2), q(w2))
    y = {"none": lambda t: t, "relu": F.relu}[act](y).float()
    wcat = torch.cat([w1.permute(0, 2, 3, 1).reshape(Cout, -1), w2.reshape(Cout, C2)], dim=1).contiguous().cuda()
    xd, x2d, bd = nhwc(x, tdt), nhwc(x2, tdt), b.cuda()
    from telescope_cam_detection_amd import _capi
    for out_f32, mode in (((1, 0), (1, 10)) if dt == "f32" else ((0, 0), (1, 0), (0, 3), (0, 10))):
        if dt == "f32" and B * H * W * ((Cout + 127) // 128) < 512 * 128:
            # fp32 launches need >= 512 tiles for the LDS-DMA kernels: the library must refuse the shape, not fall back
            yd = torch.zeros(B, H, W, Cout, dtype=torch.float32, device="cuda")
            rc = L.rtd_op_conv_dual(code, xd.data_ptr(), x2d.data_ptr(), wcat.data_ptr(), bd.data_ptr(), None, yd.data_ptr(),
                                    B, H, W, Cin, C2, Cout, k, 1, k // 2, {"none": 0, "relu": 1}[act], 0, out_f32, 0)
            assert rc != 0
            continue
        _capi.debug_option("conv_mode", mode)
        yd = torch.full((B, H, W, Cout), float("nan"), dtype=torch.float32 if out_f32 else tdt, device="cuda")
        ck(L, L.rtd_op_conv_dual(code, xd.data_ptr(), x2d.data_ptr(), wcat.data_ptr(), bd.data_ptr(), None, yd.data_ptr(),
                                 B, H, W, Cin, C2, Cout, k, 1, k // 2, {"none": 0, "relu": 1}[act], 0, out_f32, 0))
        got = yd.float().cpu().permute(0, 3, 1, 2)
        tol = dict(atol=2e-5, rtol=2e-5) if dt == "f32" else (dict(atol=3e-3, rtol=3e-3) if out_f32 else dict(atol=3e-2, rtol=1e-2))
        torch.testing.assert_close(got, y, **tol)
    _capi.debug_option("conv_mode", 0)


@pytest.mark.parametrize("seed", list(range(12)))
def test_dual_input_conv_random_shapes(L, seed):
    """ConvArgs::x2 / x_up2 on random small maps (ragged tiles, tiles across images), 1x1 and 3x3 main convs."""
    rng = np.random.default_rng(7700 + seed)
    Cin, C2 = int(rng.choice([64, 128, 256])), int(rng.choice([64, 128, 256]))
    Cout = int(rng.choice([128, 256, 320]))
    k = int(rng.choice([1, 3]))
    up2 = int(k == 1 and rng.integers(0, 2))
    B = int(rng.integers(1, 4))
    H, W = int(rng.integers(1, 20)) * 2, int(rng.integers(1, 20)) * 2
    act = str(rng.choice(["none", "relu", "silu"]))
    code, tdt = DT["bf16"]
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn(B, Cin, H // 2 if up2 else H, W // 2 if up2 else W, generator=g)
    x2 = torch.randn(B, C2, H, W, generator=g)
    w1 = torch.randn(Cout, Cin, k, k, generator=g) * (1.0 / (Cin * k * k)) ** 0.5
    w2 = torch.randn(Cout, C2, 1, 1, generator=g) * (1.0 / C2) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    q = lambda t: t.to(tdt).double()
    xin = F.interpolate(q(x), scale_factor=2.0, mode="nearest") if up2 else q(x)
    y = F.conv2d(xin, q(w1), b.double(), padding=k // 2) + F.conv2d(q(x2), q(w2))
    y = {"none": lambda t: t, "relu": F.relu, "silu": F.silu}[act](y).float()
    wcat = torch.cat([w1.permute(0, 2, 3, 1).reshape(Cout, -1), w2.reshape(Cout, C2)], dim=1).contiguous().cuda()
    xd, x2d, bd = nhwc(x, tdt), nhwc(x2, tdt), b.cuda()
    yd = torch.full((B, H, W, Cout), float("nan"), dtype=tdt, device="cuda")
    rc = L.rtd_op_conv_dual(code, xd.data_ptr(), x2d.data_ptr(), wcat.data_ptr(), bd.data_ptr(), None, yd.data_ptr(),
                            B, H, W, Cin, C2, Cout, k, 1, k // 2, {"none": 0, "relu": 1, "silu": 2}[act], 0, 0, up2)
    tiles = ((B * H * W + 127) // 128) * ((Cout + 127) // 128)
    if tiles < 4:                       # fewer tiles than the LDS-DMA kernels take: refused, never a silent fallback
        assert rc != 0
        return
    ck(L, rc)
    torch.testing.assert_close(yd.float().cpu().permute(0, 3, 1, 2), y, atol=3e-2, rtol=1e-2,
                               msg=lambda m: f"{m} shape {(B, H, W, Cin, C2, Cout, k, up2, act)}")


@pytest.mark.parametrize("seed", list(range(16)))
def test_bf16_1x1_convs_random_small_shapes(L, seed):
    """bf16 1x1 convs on random small maps - images smaller than a tile, tiles that straddle several images, ragged last tiles - with every
    activation / residual form (round 5: the bf16 streaming kernels are gone; these shapes now run on the bf16 engine's one tile path)."""
    rng = np.random.default_rng(4200 + seed)
    Cin = int(rng.choice([64, 128, 256]))
    Cout = int(rng.choice([64, 128, 256, 512]) if Cin != 256 else rng.choice([64, 128, 256, 384, 640]))
    B, H, W = int(rng.integers(1, 5)), int(rng.integers(1, 40)), int(rng.integers(1, 40))
    act = str(rng.choice(["none", "relu", "silu"]))
    res_mode = int(rng.integers(0, 3))
    code, tdt = DT["bf16"]
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * (1.0 / Cin) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    res = torch.randn(B, Cout, H, W, generator=g) if res_mode else None
    q = lambda t: t.to(tdt).double()
    y = F.conv2d(q(x), q(w), b.double())
    if res_mode == 1:
        y = y + q(res)
    y = {"none": lambda t: t, "relu": F.relu, "silu": F.silu}[act](y)
    if res_mode == 2:
        y = y + q(res)
    from telescope_cam_detection_amd import _capi
    _capi.debug_option("conv_mode", 0)
    try:
        xd, wd, bd = nhwc(x, tdt), w.permute(0, 2, 3, 1).contiguous().cuda(), b.cuda()
        rd = nhwc(res, tdt) if res is not None else None
        yd = torch.full((B, H, W, Cout), float("nan"), dtype=tdt, device="cuda")
        ck(L, L.rtd_op_conv(code, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), rd.data_ptr() if rd is not None else None, yd.data_ptr(),
                            B, H, W, Cin, Cout, 1, 1, 1, 0, {"none": 0, "relu": 1, "silu": 2}[act], res_mode, 0))
    finally:
        _capi.debug_option("conv_mode", 0)
    got = yd.float().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all(), (B, H, W, Cin, Cout, act, res_mode)
    torch.testing.assert_close(got, y.float(), atol=3e-2, rtol=1e-2, msg=lambda m: f"{m} shape {(B, H, W, Cin, Cout, act, res_mode)}")


@pytest.mark.parametrize("case", [(2, 40, 36, 256, 256, 512, "silu"), (1, 26, 30, 64, 128, 256, "none"), (3, 20, 20, 128, 64, 128, "relu")])
def test_conv_upsampled_first_input(L, case):
    """ConvArgs::x_up2: a 1x1 conv over cat([upsample2x_nearest(lat), x2]) read from the half-size `lat` and x2 directly."""
    B, H, W, Cin, C2, Cout, act = case
    code, tdt = DT["bf16"]
    g = torch.Generator().manual_seed(900 + Cin + H)
    lat = torch.randn(B, Cin, H // 2, W // 2, generator=g)
    x2 = torch.randn(B, C2, H, W, generator=g)
    w = torch.randn(Cout, Cin + C2, generator=g) * (1.0 / (Cin + C2)) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    q = lambda t: t.to(tdt).double()
    cat = torch.cat([F.interpolate(q(lat), scale_factor=2.0, mode="nearest"), q(x2)], dim=1)
    y = F.conv2d(cat, q(w)[:, :, None, None], b.double())
    y = {"none": lambda t: t, "relu": F.relu, "silu": F.silu}[act](y).float()
    ld, x2d, wd, bd = nhwc(lat, tdt), nhwc(x2, tdt), w.contiguous().cuda(), b.cuda()
    yd = torch.full((B, H, W, Cout), float("nan"), dtype=tdt, device="cuda")
    ck(L, L.rtd_op_conv_dual(code, ld.data_ptr(), x2d.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, yd.data_ptr(),
                             B, H, W, Cin, C2, Cout, 1, 1, 0, {"none": 0, "relu": 1, "silu": 2}[act], 0, 0, 1))
    torch.testing.assert_close(yd.float().cpu().permute(0, 3, 1, 2), y, atol=3e-2, rtol=1e-2)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv(L, dt, case):
    B, H, W, Cin, Cout, k, stride, pad, act, res_mode = case
    if dt == "bf16" and Cin % 8:
        pytest.skip("bf16 needs Cin % 8 == 0")
    code, tdt = DT[dt]
    g = torch.Generator().manual_seed(100 + CONV_CASES.index(case))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (1.0 / (Cin * k * k)) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, Cout, OH, OW, generator=g) if res_mode else None
    # reference on the values the kernel actually sees (inputs rounded to the storage type)
    xq, wq = x.to(tdt).float(), w.to(tdt).float()
    rq = res.to(tdt).float() if res is not None else None
    y = F.conv2d(xq.double(), wq.double(), b.double(), stride=stride, padding=pad)
    if res_mode == 1:
        y = y + rq.double()
    y = {"none": lambda t: t, "relu": F.relu, "silu": F.silu, "gelu": F.gelu}[act](y)
    if res_mode == 2:
        y = y + rq.double()
    y = y.float()
    xd = nhwc(x, tdt)
    wd = w.permute(0, 2, 3, 1).contiguous().cuda()      # OHWI fp32
    bd = b.cuda()
    rd = nhwc(res, tdt) if res is not None else None
    from telescope_cam_detection_amd import _capi
    # conv_mode 0 = auto (LDS-DMA kernel where eligible), 1 = the register-staged fallback kernel only,
    # 3 / 4 = wave-specialised LDS-DMA kernel (4 / 2 stages),
    # 10 = the 128 x 64 tile of the wave-specialised kernel (auto below 160 tiles) on every shape the LDS-DMA kernels take
    # (round 5: the 256-pixel, A-stationary, streaming and direct-3x3 bf16 generations were removed; bf16 / fp32 keep this one path)
    for out_f32, mode in (((1, 0), (1, 1), (1, 3), (1, 10)) if dt == "f32" else ((0, 0), (1, 0), (0, 1), (0, 3), (0, 4), (0, 10), (1, 10))):
        _capi.debug_option("conv_mode", mode)
        yd = torch.full((B, OH, OW, Cout), float("nan"), dtype=torch.float32 if out_f32 else tdt, device="cuda")
        ck(L, L.rtd_op_conv(code, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), rd.data_ptr() if rd is not None else None,
                            yd.data_ptr(), B, H, W, Cin, Cout, k, k, stride, pad, {"none": 0, "relu": 1, "silu": 2, "gelu": 3}[act],
                            res_mode, out_f32))
        got = yd.float().cpu().permute(0, 3, 1, 2)
        assert torch.isfinite(got).all()
        if dt == "f32":
            tol = dict(atol=2e-5, rtol=2e-5)
        elif out_f32:
            tol = dict(atol=2e-3, rtol=2e-3)       # bf16 products are exact in fp32; only accumulation order differs
        else:
            tol = dict(atol=2e-2, rtol=1e-2)       # + one bf16 rounding of the output
        torch.testing.assert_close(got, y, **tol, msg=lambda m: f"conv_mode {mode} out_f32 {out_f32} {case}: {m}")
    _capi.debug_option("conv_mode", 0)


SPLIT_CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad, act, res_mode, out_f32   (Cin % 32 == 0; F16X2 output: Cout % 32 == 0)
    (2, 20, 20, 64, 64, 1, 1, 0, "relu", 0, 0),         # 128 x 64 tile
    (1, 33, 29, 32, 96, 3, 1, 1, "silu", 0, 0),         # one channel group per tap, ragged M, Cout not a tile multiple
    (2, 40, 40, 64, 128, 3, 2, 1, "relu", 1, 0),        # stride 2, pre-activation F16X2 residual
    (1, 16, 16, 128, 256, 1, 1, 0, "none", 2, 0),       # post-activation residual
    (4, 60, 264, 32, 32, 3, 1, 1, "relu", 0, 0),        # stem.1: the direct (filter-in-registers) split kernel, ragged tiles both ways
    (4, 68, 232, 32, 64, 3, 1, 1, "silu", 0, 0),        # stem.2: two channel groups per block
    (2, 40, 40, 32, 32, 3, 1, 1, "relu", 0, 0),         # small map: stays on the tiled kernel (half-empty 64-channel tile)
    (4, 80, 80, 128, 128, 3, 1, 1, "silu", 1, 0),       # 2-stage kernel (>= 257 blocks)
    (4, 100, 97, 128, 192, 3, 1, 1, "silu", 2, 0),      # ragged M, partial last N tile, post residual
    (2, 160, 160, 64, 256, 1, 1, 0, "relu", 1, 0),      # stage-0 c3
    (1, 8400, 1, 256, 1536, 1, 1, 0, "none", 0, 1),     # value projection: fp32 output
    (1, 400, 1, 2048, 256, 1, 1, 0, "none", 0, 1),      # enc.proj.2: long K, fp32 output, 4-stage kernel
    (2, 20, 20, 512, 512, 3, 1, 1, "relu", 0, 0),       # stage-3 c2: K = 4608 real channels x taps (two-pass split-K, 2 slices, F16X2 out)
    (1, 70, 50, 256, 64, 1, 1, 0, "gelu", 0, 0),
    (2, 40, 41, 256, 1024, 1, 1, 0, "relu", 1, 1),      # fp32 output with a F16X2 residual
    (8, 40, 40, 256, 256, 3, 1, 1, "silu", 2, 0),       # 40^2 x 8 maps: flexible tile height (112 px -> 230 blocks), post residual
    (8, 80, 80, 64, 256, 3, 1, 1, "relu", 0, 0),        # 80^2 x 8 maps: 208-pixel tiles, 3 stages, one block per CU
    (8, 20, 20, 256, 512, 3, 1, 1, "relu", 1, 0),       # 20^2 x 8 maps: 64-pixel tiles
    (3, 37, 29, 512, 192, 1, 1, 0, "none", 0, 1),       # ragged M and N, fp32 output, K = 512 (16 K-steps: flexible tiles)
    (1, 81, 80, 64, 256, 1, 1, 0, "silu", 2, 0),        # streaming split kernel (>= 6400 pixels per image): ragged last 32-pixel tile, post residual
    (2, 80, 80, 128, 512, 1, 1, 0, "relu", 1, 0),       # stage-1 c3: K = 128, two 256-channel blocks in gridDim.y
    (1, 100, 70, 64, 512, 1, 1, 0, "none", 0, 0),       # no residual, linear
    (1, 80, 80, 256, 1024, 1, 1, 0, "none", 0, 0),      # streaming kernel with K = 256 (one pixel-tile buffer), four channel blocks, F16X2 out
    (1, 90, 75, 256, 1280, 1, 1, 0, "silu", 0, 1),      # ... fp32 rows out, ragged last tile
    (1, 20, 20, 512, 256, 3, 1, 1, "silu", 1, 1),       # 144 K-steps on 8 tiles per image: two-pass split-K in 4 slices, residual + fp32 rows in the reduce pass
    (2, 160, 160, 64, 64, 3, 1, 1, "relu", 0, 0),       # stage-0 c2: the direct 64-channel kernel (>= 128 8 x 16 tiles per image)
    (1, 130, 125, 64, 64, 3, 1, 1, "silu", 0, 0),       # ... ragged tiles both ways
]


@pytest.mark.parametrize("case", SPLIT_CONV_CASES)
def test_conv_split_f16x3(L, case):
    """F16X2 operands (hi/lo fp16 pairs, 32-channel groups), three MFMAs per product: against fp64 on the values the kernel sees
    (inputs rounded to hi + lo).  Error budget: the dropped lo*lo term (2^-18 per product) + fp32 accumulation + one hi/lo rounding
    of the output (2^-18) -> 2e-5 relative, 250x tighter than the bf16 kernels' tolerance."""
    from telescope_cam_detection_amd import _capi
    B, H, W, Cin, Cout, k, stride, pad, act, res_mode, out_f32 = case
    g = torch.Generator().manual_seed(4000 + SPLIT_CONV_CASES.index(case))
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (1.0 / (Cin * k * k)) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, OH, OW, Cout, generator=g) if res_mode else None
    xs = _capi.to_split(x.numpy())
    xq = torch.from_numpy(_capi.from_split(xs)).permute(0, 3, 1, 2)
    wq = torch.from_numpy(_capi.from_split(_capi.to_split(w.permute(0, 2, 3, 1).reshape(Cout, -1).numpy()))).reshape(Cout, k, k, Cin).permute(0, 3, 1, 2)
    rs = _capi.to_split(res.numpy()) if res is not None else None
    rq = torch.from_numpy(_capi.from_split(rs)).permute(0, 3, 1, 2).double() if res is not None else None
    y = F.conv2d(xq.double(), wq.double(), b.double(), stride=stride, padding=pad)
    if res_mode == 1:
        y = y + rq
    y = {"none": lambda t: t, "relu": F.relu, "silu": F.silu, "gelu": F.gelu}[act](y)
    if res_mode == 2:
        y = y + rq
    xd = torch.from_numpy(xs.view(np.int16)).cuda()
    wd = w.permute(0, 2, 3, 1).contiguous().cuda()
    bd = b.cuda()
    rd = torch.from_numpy(rs.view(np.int16)).cuda() if rs is not None else None
    if out_f32:
        yd = torch.full((B, OH, OW, Cout), float("nan"), dtype=torch.float32, device="cuda")
    else:
        yd = torch.full((B, OH, OW, 2 * Cout), -1, dtype=torch.int16, device="cuda")
    # dispatch variants: "auto" = the default; "flex" = flexible-height tiles (conv_igemm_wsf_kernel) on every grid and K length;
    # "tiled" = only the fixed-tile kernel (no streaming / direct 3x3 / flexible kernels) with 2-stage 128-wide or 4-stage 64-wide tiles everywhere
    # "quad" = the 160..256-pixel one-block-per-CU kernel (conv_igemm_wsq_kernel) wherever its shape rules allow (Cout > 64), else the tiled kernel
    tiled_bits = None
    for ws2, ws64, kern in ((257, 160, "auto"), (257, 160, "flex"), (1, 0, "tiled"), (1 << 30, 1 << 30, "tiled"), (1, 0, "quad")):
        _capi.debug_option("reset", 0)
        _capi.debug_option("split_ws2_min_blocks", ws2)
        _capi.debug_option("split_ws64_max_blocks", ws64)
        if kern != "quad":
            _capi.debug_option("split_wsq", 1 if kern == "auto" else 0)
        if kern in ("tiled", "quad"):
            _capi.debug_option("split_flex", 0)
            _capi.debug_option("split_sx", 0)        # the tiled kernel on the streaming kernel's shapes too
            _capi.debug_option("conv_reg", 0)        # ... and on the direct 3x3 kernels'
        if kern == "quad":
            _capi.debug_option("split_wsq", 2)
            _capi.debug_option("split_wsq_min_nk", 1)
            _capi.debug_option("split_k2", 0)
        if kern == "flex":
            _capi.debug_option("split_flex_small_max", 1 << 30)
            _capi.debug_option("split_flex_min_nk", 1)
        ck(L, L.rtd_op_conv(_capi.DT_F16X2, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), rd.data_ptr() if rd is not None else None,
                            yd.data_ptr(), B, H, W, Cin, Cout, k, k, stride, pad, {"none": 0, "relu": 1, "silu": 2, "gelu": 3}[act],
                            res_mode, out_f32))
        got = yd.cpu().numpy() if out_f32 else _capi.from_split(yd.cpu().numpy().view(np.uint16))
        got = torch.from_numpy(got).permute(0, 3, 1, 2).double()
        assert torch.isfinite(got).all()
        err = (got - y).abs().max().item() / y.abs().max().item()
        rel = (torch.linalg.norm(got - y) / torch.linalg.norm(y)).item()
        print(f"split conv {case}: max err / max |y| {err:.2e}, rel l2 {rel:.2e}")
        assert err < 2e-5 and rel < 1e-5, (case, ws2, ws64, kern, err, rel)
        # the tile kernels walk K in one order and issue the three products in one order: which of them ran a layer does not show in its bits
        if kern == "tiled" and ws2 == 1:
            _capi.debug_option("split_k2", 0)
            ck(L, L.rtd_op_conv(_capi.DT_F16X2, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), rd.data_ptr() if rd is not None else None,
                                yd.data_ptr(), B, H, W, Cin, Cout, k, k, stride, pad, {"none": 0, "relu": 1, "silu": 2, "gelu": 3}[act],
                                res_mode, out_f32))
            tiled_bits = yd.cpu().numpy().copy()
        if kern == "quad":
            np.testing.assert_array_equal(yd.cpu().numpy().view(np.uint32 if out_f32 else np.uint16), tiled_bits.view(np.uint32 if out_f32 else np.uint16))
    _capi.debug_option("reset", 0)


@pytest.mark.parametrize("shape", [(1, 40, 40, 64, 128, 1), (1, 160, 160, 64, 256, 1), (1, 24, 24, 64, 64, 3)])
def test_a_nan_stays_a_nan_through_the_pair_format(L, shape):
    """split2 saturates at +-65504 instead of producing inf - but a NaN must stay a NaN, on the device (epilogue of every
    pair kernel: tiled, streaming, direct 3x3 paths) as in the host mirror `_capi.to_split`: otherwise an upstream fault would be
    laundered into a finite bound and no isfinite check could see it.  An overflow still saturates."""
    from telescope_cam_detection_amd import _capi
    B, H, W, Cin, Cout, k = shape
    x = torch.ones(B, H, W, Cin) * 0.5
    x[0, 3, 5, 7] = float("nan")
    x[0, 9, 2, 1] = 3.0e38                                     # beyond fp16: saturates to 65504 in both mirrors
    xs = _capi.to_split(x.numpy())
    back = _capi.from_split(xs)
    assert np.isnan(back[0, 3, 5, 7]) and back[0, 9, 2, 1] == 65504.0 and np.isfinite(np.delete(back.reshape(-1), (3 * W + 5) * Cin + 7)).all()
    w = torch.zeros(Cout, Cin, k, k)
    for c in range(Cout):
        w[c, c % Cin, k // 2, k // 2] = 1.0                    # output channel c copies input channel c % Cin (centre tap)
    wd = w.permute(0, 2, 3, 1).contiguous().cuda()
    bd = torch.zeros(Cout).cuda()
    xd = torch.from_numpy(xs.view(np.int16)).cuda()
    yd = torch.full((B, H, W, 2 * Cout), -1, dtype=torch.int16, device="cuda")
    ck(L, L.rtd_op_conv(_capi.DT_F16X2, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, yd.data_ptr(), B, H, W, Cin, Cout, k, k, 1, k // 2, 0, 0, 0))
    got = _capi.from_split(yd.cpu().numpy().view(np.uint16))
    nan_at = np.argwhere(np.isnan(got))
    assert len(nan_at) > 0 and all(tuple(i[:3]) == (0, 3, 5) or k == 3 for i in nan_at)        # visible where it was put (3x3: also its window)
    assert np.isnan(got[0, 3, 5, 7])
    assert got[0, 9, 2, 1] == 65504.0                          # overflow saturates, finite
    assert got[0, 20, 20, 3] == 0.5


SX_NEXT_CASES = [
    # B, H, W, Cin, C2 (0 = no second input), Cnext, act, res_mode, next_act
    (2, 160, 160, 64, 64, 64, "relu", 0, "relu"),    # stage-0 block 0: c3 + folded shortcut, next block's c1 fused
    (1, 160, 160, 64, 0, 64, "relu", 1, "relu"),     # stage-0 blocks 1..: residual
    (1, 81, 80, 64, 0, 128, "relu", 1, "relu"),      # last block of stage 0 -> stage-1 block 0 c1 (128 channels), ragged last tile
    (3, 90, 90, 64, 0, 64, "silu", 2, "none"),       # tiles straddling images, post residual, linear follower
]


@pytest.mark.parametrize("case", SX_NEXT_CASES)
def test_conv_split_streaming_with_fused_next(L, case):
    """conv1x1_sx_kernel with ConvArgs::next_*: y = act(W [x | x2] + b (+ res)) and y1 = next_act(W1 y + b1) from one launch, both against
    fp64 on the values the kernel sees (y1 from the hi/lo-rounded y the kernel keeps in LDS = what a separate launch would read back)."""
    from telescope_cam_detection_amd import _capi
    B, H, W, Cin, C2, Cnext, act, res_mode, next_act = case
    Cout = 256
    g = torch.Generator().manual_seed(5200 + SX_NEXT_CASES.index(case))
    rq = lambda t: torch.from_numpy(_capi.from_split(_capi.to_split(t.numpy())))      # round to hi + lo
    x = torch.randn(B, H, W, Cin, generator=g)
    x2 = torch.randn(B, H, W, C2, generator=g) if C2 else None
    w = torch.randn(Cout, Cin + C2, generator=g) * (1.0 / (Cin + C2)) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    res = torch.randn(B, H, W, Cout, generator=g) if res_mode else None
    w1 = torch.randn(Cnext, Cout, generator=g) * (1.0 / Cout) ** 0.5
    b1 = torch.randn(Cnext, generator=g) * 0.1
    xin = rq(x) if x2 is None else torch.cat([rq(x), rq(x2)], -1)
    y = xin.double() @ rq(w).double().t() + b.double()
    if res_mode == 1:
        y = y + rq(res).double()
    y = {"none": lambda t: t, "relu": F.relu, "silu": F.silu}[act](y)
    if res_mode == 2:
        y = y + rq(res).double()
    dev = lambda t: torch.from_numpy(_capi.to_split(t.numpy()).view(np.int16)).cuda()
    xd, x2d, rd = dev(x), (dev(x2) if x2 is not None else None), (dev(res) if res is not None else None)
    wd, bd, w1d, b1d = w.contiguous().cuda(), b.cuda(), w1.contiguous().cuda(), b1.cuda()
    yd = torch.full((B, H, W, 2 * Cout), -1, dtype=torch.int16, device="cuda")
    y1d = torch.full((B, H, W, 2 * Cnext), -1, dtype=torch.int16, device="cuda")
    A = {"none": 0, "relu": 1, "silu": 2}
    ck(L, L.rtd_op_conv_next(_capi.DT_F16X2, xd.data_ptr(), x2d.data_ptr() if x2d is not None else None, wd.data_ptr(), bd.data_ptr(),
                             rd.data_ptr() if rd is not None else None, yd.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), y1d.data_ptr(),
                             B, H, W, Cin, C2, Cout, Cnext, A[act], res_mode, A[next_act]))
    got = torch.from_numpy(_capi.from_split(yd.cpu().numpy().view(np.uint16))).double()
    got1 = torch.from_numpy(_capi.from_split(y1d.cpu().numpy().view(np.uint16))).double()
    assert torch.isfinite(got).all() and torch.isfinite(got1).all()
    err = (got - y).abs().max().item() / y.abs().max().item()
    y1 = got @ rq(w1).double().t() + b1.double()                    # the follower reads the kernel's own (rounded) y
    y1 = {"none": lambda t: t, "relu": F.relu}[next_act](y1)
    err1 = (got1 - y1).abs().max().item() / y1.abs().max().item()
    print(f"sx + next {case}: y {err:.2e}, y1 {err1:.2e}")
    assert err < 2e-5 and err1 < 2e-5, (case, err, err1)
    # the same conv without the follower gives the same y bits (kernel choice is per-image; the follower only reads the tile)
    yd2 = torch.full((B, H, W, 2 * Cout), -1, dtype=torch.int16, device="cuda")
    if x2d is None:
        ck(L, L.rtd_op_conv(_capi.DT_F16X2, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), rd.data_ptr() if rd is not None else None, yd2.data_ptr(),
                            B, H, W, Cin, Cout, 1, 1, 1, 0, A[act], res_mode, 0))
    else:
        ck(L, L.rtd_op_conv_dual(_capi.DT_F16X2, xd.data_ptr(), x2d.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, yd2.data_ptr(),
                                 B, H, W, Cin, C2, Cout, 1, 1, 0, A[act], 0, 0, 0))
    assert torch.equal(yd, yd2)
    # and the separate follower launch (tiled split kernel) agrees with the fused one to rounding
    y1d2 = torch.full((B, H, W, 2 * Cnext), -1, dtype=torch.int16, device="cuda")
    ck(L, L.rtd_op_conv(_capi.DT_F16X2, yd.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), None, y1d2.data_ptr(), B, H, W, Cout, Cnext, 1, 1, 1, 0, A[next_act], 0, 0))
    sep = torch.from_numpy(_capi.from_split(y1d2.cpu().numpy().view(np.uint16))).double()
    assert (sep - got1).abs().max().item() / y1.abs().max().item() < 2e-5
    # split_sx 0: the tiled kernels take the expand conv; a fused follower is then refused, not silently dropped
    _capi.debug_option("split_sx", 0)
    try:
        rc = L.rtd_op_conv_next(_capi.DT_F16X2, xd.data_ptr(), x2d.data_ptr() if x2d is not None else None, wd.data_ptr(), bd.data_ptr(),
                                rd.data_ptr() if rd is not None else None, yd.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), y1d.data_ptr(),
                                B, H, W, Cin, C2, Cout, Cnext, A[act], res_mode, A[next_act])
        assert rc != 0
    finally:
        _capi.debug_option("reset", 0)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("dim", [64, 256, 384])
def test_layernorm(L, dt, dim):
    code, tdt = DT[dt]
    rows = 77
    g = torch.Generator().manual_seed(dim)
    x = torch.randn(rows, dim, generator=g) * 2 + 0.5
    r = torch.randn(rows, dim, generator=g)
    gam = torch.randn(dim, generator=g)
    bet = torch.randn(dim, generator=g)
    xq, rq = x.to(tdt).float(), r.to(tdt).float()
    for use_res in (False, True):
        ref = F.layer_norm(xq + (rq if use_res else 0), (dim,), gam, bet, 1e-5)
        xd, rd, gd, bd = x.to(tdt).cuda(), r.to(tdt).cuda(), gam.cuda(), bet.cuda()   # keep device buffers alive
        yd = torch.empty(rows, dim, dtype=torch.float32, device="cuda")
        ck(L, L.rtd_op_layernorm(code, xd.data_ptr(), rd.data_ptr() if use_res else None, gd.data_ptr(),
                                 bd.data_ptr(), yd.data_ptr(), rows, dim, 1))
        torch.testing.assert_close(yd.cpu(), ref, atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 400, 8, 32), (1, 300, 8, 32), (1, 130, 4, 48), (2, 70, 2, 64), (1, 1600, 8, 48), (3, 50, 2, 32)])
def test_attention(L, dt, shape):
    code, tdt = DT[dt]
    B, Lq, heads, hd = shape
    D = heads * hd
    g = torch.Generator().manual_seed(Lq)
    qk = torch.randn(B, Lq, 2 * D, generator=g)
    v = torch.randn(B, Lq, D, generator=g)
    qkq, vq = qk.to(tdt).float(), v.to(tdt).float()
    q = qkq[..., :D].view(B, Lq, heads, hd).transpose(1, 2)
    k = qkq[..., D:].view(B, Lq, heads, hd).transpose(1, 2)
    vv = vq.view(B, Lq, heads, hd).transpose(1, 2)
    a = torch.softmax(q @ k.transpose(2, 3) * hd ** -0.5, -1)
    ref = (a @ vv).transpose(1, 2).reshape(B, Lq, D)
    od = torch.empty(B, Lq, D, dtype=tdt, device="cuda")
    qkd, vd = qk.to(tdt).cuda(), v.to(tdt).cuda()
    ck(L, L.rtd_op_attention(code, qkd.data_ptr(), vd.data_ptr(), od.data_ptr(), B, Lq, heads, hd))
    tol = dict(atol=2e-5, rtol=2e-5) if dt == "f32" else dict(atol=1.5e-2, rtol=1.5e-2)
    torch.testing.assert_close(od.float().cpu(), ref, **tol)


def _msdeform_ref(value, offaw, ref, heads, hd, shapes, n_points, offset_scale):
    """HF:rt_detr_v2/modeling_rt_detr_v2.py:44-115,186-221 restated with torch ops (fp32)."""
    B, S, D = value.shape
    Q = ref.shape[1]
    Lv = len(shapes)
    LP = Lv * n_points
    off = offaw[..., : heads * LP * 2].view(B, Q, heads, LP, 2)
    aw = torch.softmax(offaw[..., heads * LP * 2:].view(B, Q, heads, LP), -1)
    r = ref[:, :, None, :]
    scale = torch.full((LP, 1), 1.0 / n_points)
    loc = r[:, :, None, :, :2] + off * scale * r[:, :, None, :, 2:] * offset_scale
    grids = (2 * loc - 1).permute(0, 2, 1, 3, 4).flatten(0, 1)
    vlist = value.view(B, S, heads, hd).permute(0, 2, 3, 1).flatten(0, 1).split([h * w for h, w in shapes], dim=-1)
    samp = []
    for l, (h, w) in enumerate(shapes):
        samp.append(F.grid_sample(vlist[l].reshape(B * heads, hd, h, w), grids[:, :, l * n_points:(l + 1) * n_points],
                                  mode="bilinear", padding_mode="zeros", align_corners=False))
    a = aw.permute(0, 2, 1, 3).reshape(B * heads, 1, Q, LP)
    return (torch.cat(samp, -1) * a).sum(-1).view(B, heads * hd, Q).transpose(1, 2).contiguous()


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_msdeform(L, dt):
    code, tdt = DT[dt]
    B, Q, heads, hd, n_points = 2, 75, 8, 32, 4
    shapes = [(20, 24), (10, 12), (5, 6)]
    S = sum(h * w for h, w in shapes)
    g = torch.Generator().manual_seed(7)
    value = torch.randn(B, S, heads * hd, generator=g)
    offaw = torch.randn(B, Q, heads * 12 * 3, generator=g) * 2.0        # large offsets: many taps leave the map
    ref = torch.rand(B, Q, 4, generator=g)
    ref[..., 2:] = ref[..., 2:] * 0.5 + 0.05
    ref[0, 0] = torch.tensor([1.0, 1.0, 1.0, 1.0])                        # the masked-anchor reference box
    want = _msdeform_ref(value.to(tdt).float(), offaw, ref, heads, hd, shapes, n_points, 0.5)
    out = torch.empty(B, Q, heads * hd, dtype=torch.float32, device="cuda")
    lv = (C.c_int32 * 6)(*[v for hw in shapes for v in hw])
    vd, od_, rd = value.to(tdt).cuda(), offaw.cuda(), ref.cuda()
    ck(L, L.rtd_op_msdeform(code, vd.data_ptr(), od_.data_ptr(), rd.data_ptr(), out.data_ptr(),
                            B, Q, heads, hd, 3, n_points, lv, heads * hd, 0.5))
    torch.testing.assert_close(out.cpu(), want, atol=3e-5, rtol=1e-4)


@pytest.mark.parametrize("case", [(3, 8400, 300), (2, 24000, 300), (1, 1000, 1000), (2, 5000, 1), (1, 33600, 300)])
def test_topk_exact(L, case):
    B, N, K = case
    g = torch.Generator().manual_seed(N + K)
    keys = torch.randn(B, N, generator=g)
    keys[:, ::7] = keys[:, 3:4]            # many exact ties, some of them at the selection boundary
    keys[0, :50] = -0.0
    keys[0, 50:100] = 0.0
    idx = torch.empty(B, K, dtype=torch.int32, device="cuda")
    val = torch.empty(B, K, dtype=torch.float32, device="cuda")
    kd = keys.cuda()
    ck(L, L.rtd_op_topk(kd.data_ptr(), B, N, K, idx.data_ptr(), val.data_ptr()))
    idx, val = idx.cpu().long(), val.cpu()
    for b in range(B):
        # stable descending order = (value desc, index asc): the kernel's documented tie rule
        order = np.lexsort((np.arange(N), -keys[b].double().numpy()))[:K]
        # -0.0 sorts below +0.0 in the kernel's total order (sign bit); lexsort on doubles treats them equal
        want_v = keys[b][torch.as_tensor(order)]
        assert torch.equal(val[b], keys[b][idx[b]]), "values must be the keys at the returned indices"
        assert len(set(idx[b].tolist())) == K
        assert torch.equal(torch.sort(val[b], descending=True).values, val[b])
        torch.testing.assert_close(val[b], want_v, atol=0, rtol=0)
        same_val = val[b][1:] == val[b][:-1]
        assert (idx[b][1:][same_val & (val[b][1:] != 0)] > idx[b][:-1][same_val & (val[b][1:] != 0)]).all(), "ties: lowest index first"


@pytest.mark.parametrize("kind", ["distinct", "sigmoid", "all_equal", "two_values", "ties_at_cut", "short"])
@pytest.mark.parametrize("N,K", [(8400, 300), (24000, 300), (33600, 300), (1500, 1024), (300, 300)])
def test_topk_fast_and_general_paths(L, kind, N, K):
    """k_topk (round 5) bounds the candidates by the K-th largest per-thread maximum and ranks them by counting; more than 1024 candidates
    (heavy ties) take the general radix path.  Both must return torch.topk's values with the documented tie rule (lowest index first):
    distinct keys and score-like keys (fast path), one or two distinct values (general path), a tie group that straddles the cut."""
    B = 3
    g = torch.Generator().manual_seed(N * 7 + K + len(kind))
    if kind == "distinct":
        keys = torch.randperm(B * N, generator=g).float().view(B, N) * 0.25 - 1000.0
    elif kind == "sigmoid":
        keys = torch.sigmoid(torch.randn(B, N, generator=g) * 2.0 - 3.0)
    elif kind == "all_equal":
        keys = torch.full((B, N), 0.3125)
    elif kind == "two_values":
        keys = torch.where(torch.rand(B, N, generator=g) < 0.5, torch.tensor(1.5), torch.tensor(-2.0))
    elif kind == "ties_at_cut":
        keys = torch.randn(B, N, generator=g)
        kth = torch.sort(keys, 1, descending=True).values[:, min(K, N) - 1:min(K, N)]
        keys = torch.where((keys - kth).abs() < 0.02, kth.expand_as(keys), keys)       # a group of exact ties around the K-th value
    else:
        keys = torch.randn(B, N, generator=g)
        keys[:, N // 2:] = -float("inf")                                              # most threads hold only -inf
    idx = torch.empty(B, K, dtype=torch.int32, device="cuda")
    val = torch.empty(B, K, dtype=torch.float32, device="cuda")
    kd = keys.cuda()
    ck(L, L.rtd_op_topk(kd.data_ptr(), B, N, K, idx.data_ptr(), val.data_ptr()))
    idx, val = idx.cpu().long(), val.cpu()
    for b in range(B):
        order = np.lexsort((np.arange(N), -keys[b].double().numpy()))[:K]               # (value desc, index asc)
        assert np.array_equal(idx[b].numpy(), order), (kind, N, K, b)
        assert torch.equal(val[b], keys[b][torch.as_tensor(order)])


@pytest.mark.parametrize("size", [((720, 1280), (640, 640)), ((1080, 1920), (640, 640)), ((100, 90), (192, 128)),
                                  ((300, 260), (192, 128)), ((640, 640), (640, 640)), ((37, 53), (64, 96))])
def test_resize_matches_pil_bit_exactly(L, size):
    from PIL import Image
    (sh, sw), (dh, dw) = size
    src = np.random.default_rng(sh * 7 + sw).integers(0, 256, (sh, sw, 3), dtype=np.uint8)     # BGR
    rgb = np.ascontiguousarray(src[:, :, ::-1])
    want = np.asarray(Image.fromarray(rgb).resize((dw, dh), Image.BILINEAR), dtype=np.uint8)
    out = torch.empty(dh, dw, 8, dtype=torch.float32, device="cuda")
    sd = torch.from_numpy(src).cuda()
    ck(L, L.rtd_op_resize(sd.data_ptr(), sh, sw, out.data_ptr(), dh, dw, 1))
    got = out.cpu().numpy()
    assert (got[:, :, 3:] == 0).all()
    np.testing.assert_array_equal(got[:, :, :3], want.astype(np.float32) / np.float32(255.0))
