"""Op-level parity of every HIP kernel against torch.nn.functional on CPU.
Integer/index work (pool routing, layout) must be exact; fp32 kernels (exact-fp32 MFMA path) within 1e-5 of the
float64 result -- 100x tighter than north_star's 1e-3, so that a kernel that loses precision cannot hide behind
ReLU-kink arguments at model level; bf16 kernels within 2e-2 of the fp32 result of bf16-rounded inputs."""
import ctypes
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from s2p_amd import ops  # noqa: E402
from s2p_amd._lib import lib, ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, EPI_ADD, EPI_MUL_ACTGRAD, EPI_STORE, chunk_elems


def nhwc(x, pitch, dtype, dev):
    N, C, H, W = x.shape
    y = torch.zeros(N, H, W, pitch)
    y[..., :C] = x.permute(0, 2, 3, 1)
    return y.to(dtype).to(dev)


def nchw(y, C):
    return y[..., :C].float().cpu().permute(0, 3, 1, 2).contiguous()


def pack_fwd(w, cin_pad, dtype, dev):          # [Co,Ci,kh,kw] -> [Co][T][Cin_pad]
    Co, Ci, kh, kw = w.shape
    p = torch.zeros(Co, kh * kw, cin_pad)
    p[..., :Ci] = w.permute(0, 2, 3, 1).reshape(Co, kh * kw, Ci)
    return p.to(dtype).to(dev)


def pack_bwd(w, cin_pad, cout_pad, dtype, dev):  # [Co,Ci,kh,kw] -> [Cin_pad][T][Cout_pad]
    Co, Ci, kh, kw = w.shape
    p = torch.zeros(cin_pad, kh * kw, cout_pad)
    p[:Ci, :, :Co] = w.permute(1, 2, 3, 0).reshape(Ci, kh * kw, Co)
    return p.to(dtype).to(dev)


def rel_err(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


TOL = {torch.float32: 1e-5, torch.bfloat16: 2e-2}

CONV_CASES = [
    # cin, cout, k, stride, pad, transposed, reflect, H, W, N
    (3, 64, 7, 1, 3, False, True, 20, 20, 2),      # stem: reflect 7x7, thin input
    (64, 128, 3, 2, 1, False, False, 20, 20, 2),   # down
    (128, 256, 3, 1, 1, False, False, 9, 7, 3),    # resblock-like, ragged size
    (256, 128, 3, 2, 1, True, False, 5, 5, 2),     # up (convT, output_padding 1)
    (64, 3, 7, 1, 3, False, True, 12, 12, 2),      # out conv, Cout=3
    (6, 64, 4, 2, 2, False, False, 21, 21, 2),     # PatchGAN first layer
    (64, 128, 4, 1, 2, False, False, 6, 6, 2),     # PatchGAN stride-1 layer
    (128, 1, 4, 1, 2, False, False, 7, 7, 2),      # PatchGAN head
    (512, 1, 4, 1, 2, False, False, 6, 5, 3),      # PatchGAN head at the real width (x-stationary Cout=1 wgrad, 64 chunks)
    (512, 1, 4, 1, 2, False, False, 13, 13, 4),    # ... at the real map size: the MFMA forward's 11 pixel tiles per image (the last one partial)
    (40, 72, 1, 1, 0, False, False, 1, 1, 64),     # linear layer as 1x1 conv (non power-of-two channels)
    (128, 192, 3, 1, 1, False, False, 21, 21, 3),  # halo-resident fast path: 2 channel slabs, tiles spanning rows and images
    (64, 160, 5, 1, 2, False, False, 17, 23, 2),   # halo-resident path, 5x5 taps, ragged Cout tile
    (64, 3, 7, 1, 3, False, True, 40, 36, 2),      # out conv at a size that takes the spatially tiled MFMA path (ragged tiles)
    (32, 2, 5, 1, 2, False, False, 33, 37, 1),     # tiled path, zero padding, Cout=2, K=5
    (64, 3, 7, 1, 3, False, True, 84, 84, 2),      # out conv, row-streaming path: 3 waves, several row bands
    (64, 3, 7, 1, 3, False, True, 9, 130, 1),      # row-streaming path: two column strips, one short band
    (64, 3, 7, 1, 3, False, False, 30, 40, 3),     # row-streaming path with zero padding
    (3, 64, 3, 1, 1, False, False, 84, 84, 2),     # VGG conv1_1: the dgrad is the row-streaming kernel over dy with flipped taps
    (3, 64, 3, 1, 1, False, False, 9, 35, 3),      # ... two waves, one short band
    (3, 200, 3, 1, 1, False, False, 21, 21, 2),    # conditioning conv 3 -> many: thin-input forward kernel, 4 channel blocks (ragged)
    (3, 64, 7, 1, 3, False, True, 84, 84, 1),      # stem at the train size: thin-input forward kernel, several tiles per wave
    (3, 64, 7, 1, 3, False, False, 30, 40, 3),     # stem-like with zero padding: weight gradient on the row-streaming kernel with exchanged operands (2 bands of 18 rows)
    (3, 64, 7, 1, 3, False, True, 25, 31, 2),      # ... reflect padding, one band of 31 -> two of 16 rows (the second short)
    (256, 128, 3, 1, 1, False, False, 19, 21, 8),  # plane-resident kernel (bf16): 8 half-slabs, XCD-aware (image, slab) order, H != W
    (64, 64, 3, 1, 1, False, False, 20, 17, 9),    # plane-resident kernel: one co slab, 340-px plane (last blocks padded), N % 8 != 0
    (192, 64, 3, 1, 1, False, False, 21, 21, 2),   # plane-resident kernel: odd number of 64-channel input slabs
    (128, 256, 3, 1, 1, False, False, 21, 20, 72), # plane-resident PAIR kernel (288 tiles > 256, even slab count): two slabs per workgroup
    (64, 64, 3, 1, 1, False, False, 84, 84, 2),    # VGG conv1_2: row bands of the generalised plane kernel (17 bands of 5 rows, the last one short)
    (64, 128, 3, 1, 1, False, False, 42, 42, 3),   # VGG conv2_1: 5 bands of 10 rows, two co slabs, N * bands % 8 != 0
    (128, 128, 3, 1, 1, False, False, 30, 37, 8),  # bands on a ragged plane (11 rows per band), N * bands % 8 == 0: XCD-aware order
    (64, 128, 3, 1, 1, False, False, 19, 21, 131), # ... one pair per image, N % 8 != 0
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(hip_device, dtype, case):
    cin, cout, k, s, p, tr, refl, H, W, N = case
    dev = hip_device
    ce = chunk_elems(dtype)
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g)
    if dtype == torch.bfloat16:      # compare against fp32 math on bf16-rounded operands
        x = x.bfloat16().float(); w = w.bfloat16().float()
    op = 1 if tr else 0
    xr = x.double().requires_grad_(True); wr = w.double().requires_grad_(True)      # float64 reference
    if tr:
        y_ref = F.conv_transpose2d(xr, wr, b.double(), stride=s, padding=p, output_padding=op)
    elif refl:
        y_ref = F.conv2d(F.pad(xr, (p, p, p, p), mode="reflect"), wr, b.double(), stride=s)
    else:
        y_ref = F.conv2d(xr, wr, b.double(), stride=s, padding=p)
    dy = torch.randn(y_ref.shape, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    y_ref.backward(dy.double())

    cin_pad, cout_pad = ops.pad_to(cin, ce), ops.pad_to(cout, ce)
    geom = ops.ConvGeom(cin, cout, k, s, p, transposed=tr, reflect=refl, output_padding=op)
    w_std = w.permute(1, 0, 2, 3) if tr else w          # [Co,Ci,kh,kw] view of the weight
    wf = pack_fwd(w_std, cin_pad, dtype, dev)
    wb = pack_bwd(w_std, cin_pad, cout_pad, dtype, dev)
    xd = nhwc(x, cin_pad, dtype, dev)
    y = ops.conv_fwd(geom, xd, wf, b.to(dev), cin_pad)
    torch.cuda.synchronize()
    assert rel_err(nchw(y, cout), y_ref.detach()) < TOL[dtype]
    if y.shape[3] > cout:      # padded output channels are written as zeros
        assert float(y[..., cout:].float().abs().max()) == 0.0

    dyd = nhwc(dy, cout_pad, dtype, dev)
    dx = ops.conv_dgrad(geom, dyd, wb, tuple(xd.shape), cin_pad)
    torch.cuda.synchronize()
    assert rel_err(nchw(dx, cin), xr.grad) < TOL[dtype]

    # wgrad -> channels-last fp32 [rows][T][cols]
    rows, cols = (cin, cout) if tr else (cout, cin)
    dw = torch.zeros(rows, k * k, cols, device=dev)
    dbf = None if tr else torch.zeros(cout, device=dev)          # bias gradient fused into the wgrad pass
    ops.conv_wgrad(geom, xd, dyd, dw, cin_pad, cin, cout, db=dbf)
    torch.cuda.synchronize()
    dw_ref = wr.grad.permute(0, 2, 3, 1).reshape(rows, k * k, cols)
    assert rel_err(dw.cpu(), dw_ref) < TOL[dtype]
    db = torch.zeros(cout, device=dev)
    ops.channel_sum(dyd, cout, db)
    assert rel_err(db.cpu(), dy.sum((0, 2, 3))) < TOL[dtype]
    if dbf is not None:
        assert rel_err(dbf.cpu(), dy.sum((0, 2, 3))) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("gshape", [(3, 16, 32), (5, 16, 224)])   # the second has a dY pitch > 1024: register-staged wgrad + fused db
def test_conv_groups_epilogues(hip_device, dtype, gshape):
    dev = hip_device
    ce = chunk_elems(dtype)
    g = torch.Generator().manual_seed(5)
    (G, cin, cout), (N, H, W) = gshape, (2, 6, 5)
    x = torch.randn(N, G * cin, H, W, generator=g)
    w = torch.randn(G * cout, cin, 3, 3, generator=g) / 12
    b = torch.randn(G * cout, generator=g)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float(); w = w.bfloat16().float()
    xr = x.double().requires_grad_(True); wr = w.double().requires_grad_(True)       # float64 reference
    y_ref = F.leaky_relu(F.conv2d(xr, wr, b.double(), padding=1, groups=G), 0.2)
    dy = torch.randn(y_ref.shape, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    geom = ops.ConvGeom(cin, cout, 3, 1, 1, groups=G, x_gstride=cin, y_gstride=cout)
    wf = torch.stack([pack_fwd(w[i * cout:(i + 1) * cout], cin, dtype, dev) for i in range(G)])
    wb = torch.stack([pack_bwd(w[i * cout:(i + 1) * cout], cin, cout, dtype, dev) for i in range(G)])
    xd = nhwc(x, G * cin, dtype, dev)
    y = ops.conv_fwd(geom, xd, wf, b.to(dev), cin, y_pitch=G * cout, act=ACT_LRELU, slope=0.2)
    torch.cuda.synchronize()
    assert rel_err(nchw(y, G * cout), y_ref.detach()) < TOL[dtype]
    # backward through the fused activation: dpre = dy * lrelu'(y) (s2p_act_bwd), then grouped dgrad / wgrad
    y_ref.backward(dy.double())
    dyd = nhwc(dy, G * cout, dtype, dev)
    dpre = ops.act_bwd(dyd, y, ACT_LRELU, 0.2)
    dx = ops.conv_dgrad(geom, dpre, wb, tuple(xd.shape), cin)
    torch.cuda.synchronize()
    assert rel_err(nchw(dx, G * cin), xr.grad) < TOL[dtype]
    dw = torch.zeros(G, cout, 9, cin, device=dev)
    dbg = torch.zeros(G * cout, device=dev)
    ops.conv_wgrad(geom, xd, dpre, dw, cin, cin, cout, dw_gstride=cout * 9 * cin, db=dbg)
    torch.cuda.synchronize()
    assert rel_err(dbg.cpu(), nchw(dpre, G * cout).sum((0, 2, 3))) < TOL[dtype]
    assert rel_err(dw.cpu().reshape(G * cout, 9, cin), wr.grad.permute(0, 2, 3, 1).reshape(G * cout, 9, cin)) < TOL[dtype]
    # residual epilogue and fused producer-activation gradient epilogue
    geom1 = ops.ConvGeom(G * cin, G * cin, 3, 1, 1)
    w1 = torch.randn(G * cin, G * cin, 3, 3, generator=g) / 20
    if dtype == torch.bfloat16:
        w1 = w1.bfloat16().float()
    wf1 = pack_fwd(w1, G * cin, dtype, dev)
    y1 = ops.conv_fwd(geom1, xd, wf1, None, G * cin, aux=xd, epi=EPI_ADD)
    assert rel_err(nchw(y1, G * cin), x.double() + F.conv2d(x.double(), w1.double(), padding=1)) < TOL[dtype]
    a = F.relu(x)
    ad = nhwc(a, G * cin, dtype, dev)
    wb1 = pack_bwd(w1, G * cin, G * cin, dtype, dev)
    dy1 = torch.randn(N, G * cin, H, W, generator=g)
    if dtype == torch.bfloat16:
        dy1 = dy1.bfloat16().float()
    dxm = ops.conv_dgrad(geom1, nhwc(dy1, G * cin, dtype, dev), wb1, tuple(xd.shape), G * cin, aux=ad,
                         epi=EPI_MUL_ACTGRAD, aux_act=ACT_RELU)
    ref = F.conv_transpose2d(dy1.double(), w1.double(), padding=1) * (a > 0).double()
    assert rel_err(nchw(dxm, G * cin), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("modulated", [False, True])
def test_instance_norm_mat(hip_device, dtype, modulated):
    dev = hip_device
    g = torch.Generator().manual_seed(11)
    N, C, H, W = 3, 72 if dtype == torch.bfloat16 else 68, 9, 7
    x = torch.randn(N, C, H, W, generator=g) * 2 + 0.5
    gam = torch.randn(N, C, H, W, generator=g) * 0.5
    bet = torch.randn(N, C, H, W, generator=g) * 0.5
    st = torch.randn(N, 2 * C, generator=g) * 0.5
    da = torch.randn(N, C, H, W, generator=g)
    if dtype == torch.bfloat16:
        x, gam, bet, da = [t.bfloat16().float() for t in (x, gam, bet, da)]
    xr, gr, br, sr = [t.double().requires_grad_(True) for t in (x, gam, bet, st)]       # float64 reference
    if modulated:
        y_ref = F.leaky_relu(F.instance_norm(xr, eps=1e-5) * (1 + gr + sr[:, :C, None, None]) + br + sr[:, C:, None, None], 0.2)
        act = ACT_LRELU
    else:
        y_ref = F.relu(F.instance_norm(xr, eps=1e-5))
        act = ACT_RELU
    y_ref.backward(da.double())
    xd = nhwc(x, C, dtype, dev)
    gb = torch.cat([nhwc(gam, C, dtype, dev), nhwc(bet, C, dtype, dev)], 3).contiguous() if modulated else None
    std = st.to(dev) if modulated else None
    stats = ops.in_stats(xd, C)
    y = ops.in_apply_fwd(xd, C, stats, gb, 0, std, 0, act, 0.2)
    torch.cuda.synchronize()
    tol = TOL[dtype]
    assert rel_err(nchw(y, C), y_ref.detach()) < tol
    dgb = torch.empty_like(gb) if modulated else None
    dst = torch.full((N, 2 * C + 8), 7.0, device=dev) if modulated else None       # state-affine gradient, written at column 4
    dx = ops.in_bwd(nhwc(da, C, dtype, dev), xd, C, stats, gb, 0, std, 0, act, 0.2, dgb, 0, dst, 4)
    torch.cuda.synchronize()
    assert rel_err(nchw(dx, C), xr.grad) < tol * 2
    if modulated:
        assert rel_err(nchw(dgb, C), gr.grad) < tol
        assert rel_err(nchw(dgb[..., C:], C), br.grad) < tol
        assert rel_err(dst[:, 4:4 + 2 * C].cpu(), sr.grad) < tol
        assert float(dst[:, :4].min()) == 7.0 and float(dst[:, 4 + 2 * C:].min()) == 7.0      # nothing else touched
    # the one-call form (fused statistics + apply launch for small planes) and its statistics buffer fed to the backward
    y3, stats3 = ops.in_norm_fwd(xd, C, gb, 0, std, 0, act, 0.2)
    assert rel_err(nchw(y3, C), y_ref.detach()) < tol
    dx3 = ops.in_bwd(nhwc(da, C, dtype, dev), xd, C, stats3, gb, 0, std, 0, act, 0.2, dgb, 0, dst, 4)
    assert rel_err(nchw(dx3, C), xr.grad) < tol * 2
    y4, stats4 = ops.in_norm_fwd(xd, C, gb, 0, std, 0, act, 0.2)
    assert torch.equal(y3, y4) and torch.equal(stats3, stats4)
    # no atomics anywhere in the norm kernels: a second run is bitwise identical
    stats2 = ops.in_stats(xd, C)
    y2 = ops.in_apply_fwd(xd, C, stats2, gb, 0, std, 0, act, 0.2)
    dx2 = ops.in_bwd(nhwc(da, C, dtype, dev), xd, C, stats2, gb, 0, std, 0, act, 0.2, dgb, 0, dst, 4)
    assert torch.equal(stats, stats2) and torch.equal(y, y2) and torch.equal(dx, dx2)


@pytest.mark.parametrize("dtype,shape", [(torch.bfloat16, (5, 256, 7, 7)), (torch.bfloat16, (3, 512, 8, 8)), (torch.bfloat16, (4, 256, 12, 12)),
                                         (torch.bfloat16, (3, 512, 13, 13)), (torch.bfloat16, (2, 72, 16, 16)), (torch.bfloat16, (2, 64, 16, 17)),
                                         (torch.bfloat16, (3, 128, 22, 22)), (torch.bfloat16, (2, 64, 22, 23)),
                                         (torch.bfloat16, (16, 256, 42, 42)), (torch.bfloat16, (16, 64, 84, 84)),     # large planes, register-resident forward:
                                         (torch.bfloat16, (33, 128, 41, 43)), (torch.bfloat16, (20, 64, 60, 70)),     # 64- / 16-channel slabs, ragged sizes
                                         (torch.float32, (3, 132, 8, 16)), (torch.float32, (2, 64, 11, 12)), (torch.float32, (2, 68, 16, 16))])
def test_instance_norm_small_planes(hip_device, dtype, shape):
    """The PatchGAN maps (7x7 .. 13x13, plain InstanceNorm + LeakyReLU): planes of <= 256 (bf16) / 128 (fp32) pixels take the
    256-thread form of the fused forward / backward kernels, up to twice that the 512-thread form, 22x23 the 1024-thread form; the
    encoder / decoder planes (42x42 x 128, 84x84 x 64 channels) the large-plane form of the forward (28 chunks per thread).  Against
    float64, and against the two-kernel path (statistics + apply) on the same inputs."""
    dev = hip_device
    g = torch.Generator().manual_seed(23)
    N, C, H, W = shape
    x = torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3
    da = torch.randn(N, C, H, W, generator=g)
    if dtype == torch.bfloat16:
        x, da = x.bfloat16().float(), da.bfloat16().float()
    xr = x.double().requires_grad_(True)
    y_ref = F.leaky_relu(F.instance_norm(xr, eps=1e-5), 0.2)
    y_ref.backward(da.double())
    xd, dad = nhwc(x, C, dtype, dev), nhwc(da, C, dtype, dev)
    y, stats = ops.in_norm_fwd(xd, C, act=ACT_LRELU, slope=0.2)
    dx = ops.in_bwd(dad, xd, C, stats, act=ACT_LRELU, slope=0.2)
    torch.cuda.synchronize()
    tol = TOL[dtype]
    assert rel_err(nchw(y, C), y_ref.detach()) < tol
    assert rel_err(nchw(dx, C), xr.grad) < tol * 2
    stats2 = ops.in_stats(xd, C)
    y2 = ops.in_apply_fwd(xd, C, stats2, act=ACT_LRELU, slope=0.2)
    assert rel_err(y.float().cpu(), y2.float().cpu().double()) < (1e-5 if dtype == torch.float32 else 1e-2)
    y3, stats3 = ops.in_norm_fwd(xd, C, act=ACT_LRELU, slope=0.2)
    dx3 = ops.in_bwd(dad, xd, C, stats3, act=ACT_LRELU, slope=0.2)
    nst = 4 + N * C * 2             # header + one {mean, M2} pair per (image, channel): what the fused launch writes of the buffer
    assert torch.equal(y, y3) and torch.equal(stats[:nst], stats3[:nst]) and torch.equal(dx, dx3)          # no atomics: same bits again


@pytest.mark.parametrize("shape", [(2, 64, 21, 21), (1, 64, 84, 84), (3, 68, 9, 7), (2, 132, 16, 16)])
def test_instance_norm_large_mean(hip_device, shape):
    """|mean| / std = 1e3 (a near-constant, strongly biased channel): single-pass raw moments E[x^2] - E[x]^2 lose all
    digits here; the pivot-shifted partial moments + Chan merge must not.  fp32, against float64, 1e-3 (the input's own
    fp32 quantisation at 1e3 is 6e-5 of a standard deviation).  No activation here: with |xhat| known only to ~1e-4 a
    handful of LeakyReLU branches would differ from float64 and dominate a max-norm comparison of dx."""
    dev = hip_device
    g = torch.Generator().manual_seed(17)
    N, C, H, W = shape
    x = torch.randn(N, C, H, W, generator=g) + 1000.0 * torch.sign(torch.randn(1, C, 1, 1, generator=g))
    da = torch.randn(N, C, H, W, generator=g)
    xr = x.double().requires_grad_(True)
    y_ref = F.instance_norm(xr, eps=1e-5)
    y_ref.backward(da.double())
    xd = nhwc(x, C, torch.float32, dev)
    stats = ops.in_stats(xd, C)
    y = ops.in_apply_fwd(xd, C, stats, act=ACT_NONE)
    dx = ops.in_bwd(nhwc(da, C, torch.float32, dev), xd, C, stats, act=ACT_NONE)
    torch.cuda.synchronize()
    assert rel_err(nchw(y, C), y_ref.detach()) < 1e-3
    assert rel_err(nchw(dx, C), xr.grad) < 2e-3
    y3, stats3 = ops.in_norm_fwd(xd, C, act=ACT_NONE)                       # fused launch when the plane has <= 256 pixels
    dx3 = ops.in_bwd(nhwc(da, C, torch.float32, dev), xd, C, stats3, act=ACT_NONE)
    assert rel_err(nchw(y3, C), y_ref.detach()) < 1e-3 and rel_err(nchw(dx3, C), xr.grad) < 2e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pool_resize_layout(hip_device, dtype):
    dev = hip_device
    ce = chunk_elems(dtype)
    g = torch.Generator().manual_seed(3)
    N, C, H, W = 2, 16, 11, 9
    x = torch.randn(N, C, H, W, generator=g)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    xd = nhwc(x, C, dtype, dev)
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    # avg pool
    xr = x.clone().requires_grad_(True)
    yr = F.avg_pool2d(xr, 3, 2, 1, count_include_pad=False)
    dy = torch.randn(yr.shape, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    yr.backward(dy)
    y = ops.avgpool_fwd(xd)
    assert rel_err(nchw(y, C), yr.detach()) < tol
    dx = ops.avgpool_bwd(nhwc(dy, C, dtype, dev), tuple(xd.shape))
    assert rel_err(nchw(dx, C), xr.grad) < tol
    if dtype == torch.float32:
        # a channel count that is not a whole number of 16-byte chunks takes the per-element form of the two kernels (straight
        # through the C ABI: ops.* always passes a chunk-padded pitch)
        from s2p_amd._lib import check, dtype_id, lib, ptr, stream
        x5 = torch.randn(N, 5, H, W, generator=g)
        x5r = x5.clone().requires_grad_(True)
        y5r = F.avg_pool2d(x5r, 3, 2, 1, count_include_pad=False)
        dy5 = torch.randn(y5r.shape, generator=g)
        y5r.backward(dy5)
        x5d = x5.permute(0, 2, 3, 1).contiguous().to(dev)
        y5 = torch.empty(N, y5r.shape[2], y5r.shape[3], 5, device=dev)
        check(lib().s2p_avgpool3x3s2_fwd(dtype_id(dtype), ptr(x5d), N, H, W, 5, ptr(y5), stream()), "avgpool fwd, C = 5")
        assert rel_err(y5.permute(0, 3, 1, 2).cpu(), y5r.detach()) < tol
        dx5 = torch.empty_like(x5d)
        dy5d = dy5.permute(0, 2, 3, 1).contiguous().to(dev)
        check(lib().s2p_avgpool3x3s2_bwd(dtype_id(dtype), ptr(dy5d), N, H, W, 5, ptr(dx5), 0, stream()), "avgpool bwd, C = 5")
        assert rel_err(dx5.permute(0, 3, 1, 2).cpu(), x5r.grad) < tol
    # max pool (+ fused relu mask): x is a relu output
    a = F.relu(x)
    ar = a.clone().requires_grad_(True)
    mr = F.max_pool2d(F.relu(ar), 2, 2)
    dm = torch.randn(mr.shape, generator=g)
    if dtype == torch.bfloat16:
        dm = dm.bfloat16().float()
    mr.backward(dm)
    ad = nhwc(a, C, dtype, dev)
    m = ops.maxpool_fwd(ad)
    assert torch.equal(nchw(m, C), mr.detach())
    dxa = ops.maxpool_bwd(nhwc(dm, C, dtype, dev), ad)
    assert torch.equal(nchw(dxa, C), ar.grad)
    # nearest resize (exact)
    r = ops.resize_nearest(xd, 5, 4)
    assert torch.equal(nchw(r, C), F.interpolate(x, size=(5, 4), mode="nearest"))
    # layout round trip
    img = torch.randn(N, 3, H, W, generator=g)
    d = ops.nchw_to_nhwc(img.to(dev), dtype, ce)
    assert float(d[..., 3:].float().abs().max()) == 0.0
    back = ops.nhwc_to_nchw(d, 3).cpu()
    assert rel_err(back, img) < (1e-7 if dtype == torch.float32 else 1e-2)
    # reflect pad adjoint
    pr = x.clone().requires_grad_(True)
    pp = F.pad(pr, (3, 3, 3, 3), mode="reflect")
    dpp = torch.randn(pp.shape, generator=g)
    if dtype == torch.bfloat16:
        dpp = dpp.bfloat16().float()
    pp.backward(dpp)
    from s2p_amd._lib import check, dtype_id, lib, ptr, stream
    out = torch.empty_like(xd)
    dppd = nhwc(dpp, C, dtype, dev)
    check(lib().s2p_reflect_pad_bwd(dtype_id(dtype), ptr(dppd), N, H, W, C, 3, ptr(out), stream()), "fold")
    assert rel_err(nchw(out, C), pr.grad) < tol


def test_posenc_losses_adam(hip_device):
    dev = hip_device
    import s2p_oracle as O
    g = torch.Generator().manual_seed(9)
    s = torch.randn(5, 17, generator=g)
    pe = ops.posenc(s.to(dev), 10, 360).cpu()
    ref = O.positional_encoding(s, 10)
    assert rel_err(pe[:, :357], ref) < 1e-5 and float(pe[:, 357:].abs().max()) == 0
    a = torch.randn(4, 6, 6, 8, generator=g); b = torch.randn(4, 6, 6, 8, generator=g)
    loss = torch.zeros(1, device=dev)
    ga = torch.empty_like(a, device=dev)
    ops.l1_loss(a.to(dev), b.to(dev), 0.25, loss, ga)
    assert abs(float(loss) - 0.25 * float((a - b).abs().sum())) < 1e-3
    assert torch.equal(ga.cpu(), 0.25 * torch.sign(a - b))
    # several L1 terms in one launch (s2p_l1_loss_multi): different sizes, one job without a gradient, shared loss slot
    shapes = [(2, 5, 5, 64), (2, 3, 3, 256), (1, 9, 7, 8)]
    As = [torch.randn(sh, generator=g).bfloat16() for sh in shapes]; Bs = [torch.randn(sh, generator=g).bfloat16() for sh in shapes]
    lm = torch.zeros(2, device=dev)
    Gs = [torch.empty(sh, dtype=torch.bfloat16, device=dev) for sh in shapes[:2]] + [None]
    dA, dB = [t.to(dev) for t in As], [t.to(dev) for t in Bs]
    ops.l1_loss_multi([(dA[0], dB[0], 0.5, lm[0:1], Gs[0]), (dA[1], dB[1], 0.25, lm[0:1], Gs[1]), (dA[2], dB[2], 2.0, lm[1:2], None)])
    want0 = 0.5 * float((As[0].float() - Bs[0].float()).abs().sum()) + 0.25 * float((As[1].float() - Bs[1].float()).abs().sum())
    assert float(lm[0]) == pytest.approx(want0, rel=1e-5) and float(lm[1]) == pytest.approx(2.0 * float((As[2].float() - Bs[2].float()).abs().sum()), rel=1e-5)
    assert torch.equal(Gs[0].float().cpu(), 0.5 * torch.sign(As[0].float() - Bs[0].float()))
    assert torch.equal(Gs[1].float().cpu(), 0.25 * torch.sign(As[1].float() - Bs[1].float()))
    x = torch.randn(300, generator=g)
    for mode, fn in ((0, lambda v: F.relu(1 + v).sum()), (1, lambda v: F.relu(1 - v).sum()), (2, lambda v: -v.sum())):
        xr = x.clone().requires_grad_(True)
        (fn(xr) * 0.5).backward()
        l = torch.zeros(1, device=dev); gx = torch.empty(300, device=dev)
        ops.hinge_loss(x.to(dev), 300, mode, 0.5, l, gx)
        assert float(l) == pytest.approx(0.5 * float(fn(x)), rel=1e-5, abs=1e-3), (mode, float(l))
        assert torch.allclose(gx.cpu(), xr.grad)
        # the same straight on an NHWC logit map (channel 0 of a pitch-8 bf16 tensor): one launch, gradient in place
        xm = torch.zeros(3, 10, 10, 8); xm[..., 0] = x.bfloat16().float().reshape(3, 10, 10); xm[..., 1:] = 7.0
        l2 = torch.zeros(1, device=dev)
        gm = ops.hinge_loss_nhwc(xm.bfloat16().to(dev), mode, 0.5, l2)
        xq = x.bfloat16().float().requires_grad_(True)
        (fn(xq) * 0.5).backward()
        assert float(l2) == pytest.approx(0.5 * float(fn(xq.detach())), rel=1e-5, abs=1e-3)
        assert torch.equal(gm[..., 0].float().cpu().reshape(-1), xq.grad.bfloat16().float())
        assert float(gm[..., 1:].float().abs().max()) == 0.0
    # Adam vs torch.optim.Adam
    p = torch.randn(1003, generator=g); gr = torch.randn(1003, generator=g)
    pt = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=1e-3, betas=(0.0, 0.9), eps=1e-8)
    pd, m, v = p.to(dev), torch.zeros(1003, device=dev), torch.zeros(1003, device=dev)
    for step in range(1, 4):
        pt.grad = gr.clone() * step
        opt.step()
        ops.adam_step(pd, (gr * step).to(dev), m, v, 1e-3, 0.0, 0.9, 1e-8, step)
    assert rel_err(pd.cpu(), pt.detach()) < 1e-5
    # the graph-capturable form (step counter on the device; hardware sqrt / reciprocal, bias corrections once per workgroup):
    # same reference, both beta settings of the trainer (TTUR: beta1 = 0), a ragged tail (n % 4 != 0), the two-part call
    # (tail first with the tick, head second without: Pix2PixTrainer.EARLY_ADAM)
    for betas in ((0.0, 0.9), (0.5, 0.999)):
        n = 70003
        p = torch.randn(n, generator=g); gr = torch.randn(n, generator=g)
        pt = p.clone().requires_grad_(True)
        opt = torch.optim.Adam([pt], lr=2e-4, betas=betas, eps=1e-8)
        pd, m, v = p.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        sd = torch.zeros(1, dtype=torch.int32, device=dev)
        cut = 40000                       # (a multiple of 4: the head's 16-byte groups stay aligned)
        for step in range(1, 6):
            gs = (gr * (1.0 + 0.1 * step)).to(dev)
            pt.grad = gr.clone() * (1.0 + 0.1 * step)
            opt.step()
            if step % 2:
                ops.adam_step_dev(pd, gs, m, v, 2e-4, betas[0], betas[1], 1e-8, sd)
            else:
                ops.adam_step_dev_part(pd[cut:], gs[cut:], m[cut:], v[cut:], 2e-4, betas[0], betas[1], 1e-8, sd, tick=True)
                ops.adam_step_dev_part(pd[:cut], gs[:cut], m[:cut], v[:cut], 2e-4, betas[0], betas[1], 1e-8, sd, tick=False)
        assert int(sd.item()) == 5
        assert rel_err(pd.cpu(), pt.detach()) < 1e-6
        assert float((pd.cpu() - pt.detach()).abs().max()) < 1e-6          # (a few ulp of parameters of size ~1-4)


@pytest.mark.gpu
def test_leaky_relu_slope_above_one_is_rejected(hip_device):
    """The forward kernels evaluate LeakyReLU as max(v, v * slope) (one multiply, one maximum): exact for slopes <= 1, so the entry
    points refuse a larger one instead of computing something else (include/s2p_hip.h)."""
    dev = hip_device
    x = torch.randn(2, 9, 9, 64, device=dev).bfloat16()
    with pytest.raises(RuntimeError, match="slope"):
        ops.in_norm_fwd(x, 64, act=ops.ACT_LRELU, slope=1.5)
    y, _ = ops.in_norm_fwd(x, 64, act=ops.ACT_LRELU, slope=1.0)        # the boundary is allowed (and is the identity)
    y0, _ = ops.in_norm_fwd(x, 64, act=ops.ACT_NONE)
    assert torch.equal(y, y0)
    geom = ops.ConvGeom(64, 64, 3, 1, 1)
    w = torch.randn(64, 9, 64, device=dev).bfloat16()
    with pytest.raises(RuntimeError, match="slope"):
        ops.conv_fwd(geom, x, w, None, 64, act=ops.ACT_LRELU, slope=2.0)


@pytest.mark.parametrize("case", [
    # n_jobs, cin, cout, H, W, N, grouped
    (3, 64, 128, 9, 7, 3, False),       # ragged grid, several K blocks, pad rows / columns crossing block boundaries
    (2, 128, 64, 21, 21, 2, False),     # the ResBlk grid
    (4, 64, 64, 5, 5, 1, True),         # one job per group of a grouped conv (channel-offset pointers, wide pitches)
    (1, 64, 64, 30, 36, 2, False),      # wider image: 192-row X window variant
])
def test_conv_wgrad_batched_slab(hip_device, case):
    """s2p_conv2d_wgrad_batched (csrc/wgrad_slab.hip): padded-raster slab kernel + fixed-order partial reduction against
    float64 autograd, accumulation into a non-zero dw / db, and bitwise reproducibility (no atomics)."""
    nj, cin, cout, H, W, N, grouped = case
    dev = hip_device
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(7 + nj)
    geom = ops.ConvGeom(cin, cout, 3, 1, 1)
    xs = torch.randn(N, nj * cin, H, W, generator=g).bfloat16().float()
    dys = torch.randn(N, nj * cout, H, W, generator=g).bfloat16().float()
    if grouped:
        xd = nhwc(xs, nj * cin, dtype, dev); dyd = nhwc(dys, nj * cout, dtype, dev)
    dw0 = torch.randn(nj, cout, 9, cin, generator=g)
    db0 = torch.randn(nj, cout, generator=g)

    def run():
        dw = dw0.clone().to(dev); db = db0.clone().to(dev)
        jobs = []
        keep = []
        for j in range(nj):
            if grouped:
                jobs.append((xd, j * cin, dyd, j * cout, dw[j], db[j]))
            else:
                xj = nhwc(xs[:, j * cin:(j + 1) * cin], cin, dtype, dev); dyj = nhwc(dys[:, j * cout:(j + 1) * cout], cout, dtype, dev)
                keep += [xj, dyj]
                jobs.append((xj, 0, dyj, 0, dw[j], db[j] if j != 1 else None))     # one job without a bias gradient
        ops.conv_wgrad_batched(geom, jobs, cin, cin, cout)
        torch.cuda.synchronize()
        return dw.cpu(), db.cpu()

    dw, db = run()
    for j in range(nj):
        xr = xs[:, j * cin:(j + 1) * cin].double()
        wr = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv2d(xr, wr, padding=1).backward(dys[:, j * cout:(j + 1) * cout].double())
        ref = dw0[j].double() + wr.grad.permute(0, 2, 3, 1).reshape(cout, 9, cin)
        assert rel_err(dw[j], ref) < 1e-5, j                       # bf16 operands are exact in fp32: only fp32 accumulation error
        if grouped or j != 1:
            assert rel_err(db[j], db0[j].double() + dys[:, j * cout:(j + 1) * cout].double().sum((0, 2, 3))) < 1e-5, j
        else:
            assert torch.equal(db[j], db0[j])
    dw2, db2 = run()
    assert torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.parametrize("case", [(512, 512, 3, 1, 5, 5, 8), (256, 256, 3, 1, 4, 6, 6), (512, 128, 2, 1, 9, 8, 5)])
def test_conv_small_map_splitk(hip_device, case):
    """Small maps with a long K that the plane-resident kernels do not take (VGG conv5_1 on 5x5 maps; since round 4 the PatchGAN
    256->512 4x4 layers run on csrc/conv_planeg.hip): the launch cannot fill the chip, so K is split over
    blockIdx.z and a second kernel applies bias / activation / epilogue to the fixed-order sum
    (s2p_conv2d_{fwd,dgrad}_ws).  Checks the plain, residual-add and producer-activation-gradient epilogues against float64, that the
    split path is the one taken, and bitwise reproducibility."""
    cin, cout, k, p, H, W, N = case
    dev, dtype = hip_device, torch.bfloat16
    g = torch.Generator().manual_seed(3 + cin + k)
    x = torch.randn(N, cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).bfloat16().float()
    b = torch.randn(cout, generator=g)
    geom = ops.ConvGeom(cin, cout, k, 1, p)
    Ho, Wo = geom.out_hw(H, W)
    d = geom.desc(dtype, N, H, W, cin, cin, cout)
    assert lib().s2p_conv2d_fwd_workspace(ctypes.byref(d), EPI_STORE) > 0
    if cout >= 256:
        assert lib().s2p_conv2d_dgrad_workspace(ctypes.byref(d)) > 0
    xd = nhwc(x, cin, dtype, dev)
    wf, wb = pack_fwd(w, cin, dtype, dev), pack_bwd(w, cin, cout, dtype, dev)
    y_ref = F.conv2d(x.double(), w.double(), b.double(), padding=p)
    y = ops.conv_fwd(geom, xd, wf, b.to(dev), cin, act=ACT_LRELU, slope=0.2)
    torch.cuda.synchronize()
    assert rel_err(nchw(y, cout), F.leaky_relu(y_ref, 0.2)) < TOL[dtype]
    assert torch.equal(y, ops.conv_fwd(geom, xd, wf, b.to(dev), cin, act=ACT_LRELU, slope=0.2))
    # dgrad with the fused producer-activation gradient and a second incoming gradient: dx = (dgrad(dy) + g2) * relu'(a)
    dy = torch.randn(N, cout, Ho, Wo, generator=g).bfloat16().float()
    a_in = F.relu(torch.randn(N, cin, H, W, generator=g)).bfloat16().float()
    g2 = torch.randn(N, cin, H, W, generator=g).bfloat16().float()
    dyd = nhwc(dy, cout, dtype, dev)
    dx_ref = torch.nn.grad.conv2d_input((N, cin, H, W), w.double(), dy.double(), padding=p)
    dx = ops.conv_dgrad(geom, dyd, wb, tuple(xd.shape), cin)
    torch.cuda.synchronize()
    assert rel_err(nchw(dx, cin), dx_ref) < TOL[dtype]
    dx2 = ops.conv_dgrad(geom, dyd, wb, tuple(xd.shape), cin, aux=nhwc(a_in, cin, dtype, dev), epi=EPI_MUL_ACTGRAD,
                         aux_act=ACT_RELU, aux2=nhwc(g2, cin, dtype, dev))
    torch.cuda.synchronize()
    assert rel_err(nchw(dx2, cin), (dx_ref + g2.double()) * (a_in > 0).double()) < TOL[dtype]
    dx3 = ops.conv_dgrad(geom, dyd, wb, tuple(xd.shape), cin, aux=nhwc(g2, cin, dtype, dev), epi=EPI_ADD)      # dx = dgrad(dy) + g2
    torch.cuda.synchronize()
    assert rel_err(nchw(dx3, cin), dx_ref + g2.double()) < TOL[dtype]
    assert torch.equal(dx, ops.conv_dgrad(geom, dyd, wb, tuple(xd.shape), cin))


@pytest.mark.parametrize("case", [
    # cin, cout, k, stride, pad, transposed, H, W, N
    (6, 64, 4, 2, 2, False, 42, 42, 8),        # PatchGAN first layer: one tile, hundreds of split units (16-segment reduce)
    (256, 512, 4, 1, 2, False, 7, 7, 8),       # PatchGAN 256->512: 128 tiles, 8 units
    (128, 64, 3, 2, 1, True, 12, 12, 4),       # up conv (transposed form, no bias gradient)
    (64, 128, 3, 2, 1, False, 20, 20, 4),      # down conv
    # the padded-raster kernel of the 4x4 layers (csrc/wgrad_slabg.hip) at the sizes of the train step
    (64, 128, 3, 2, 1, False, 84, 84, 2),      # encoder at the real size (implicit GEMM: the 3x3 stride-2 layers stay there)
    (256, 128, 3, 2, 1, True, 21, 21, 3),      # decoder (transposed form) at the real size
    (64, 128, 4, 2, 2, False, 43, 43, 3),      # PatchGAN 4x4 stride 2 on an odd size: four classes of 4 taps, parity-1 planes one short
    (128, 256, 4, 2, 2, False, 21, 23, 2),     # ... H != W
    (256, 512, 4, 1, 2, False, 12, 12, 3),     # PatchGAN 4x4 stride 1, pad 2: two classes of 8 taps, dY one row / column larger than x
    (128, 192, 4, 2, 2, False, 9, 13, 5),      # ... three co tiles x two ci slabs per class, a raster of 6 x 8 positions, ragged split
    (64, 64, 4, 1, 2, False, 5, 9, 7),         # ... one tile per class, fewer raster blocks than the split target
])
def test_conv_wgrad_split_units_deterministic(hip_device, case):
    """s2p_conv2d_wgrad_ws: the K-split units of the LDS-DMA weight-gradient kernel store partial tiles and a second
    kernel adds them in unit order -- float64 parity, accumulation into a non-zero dw / db, bitwise reproducibility."""
    cin, cout, k, s, p, tr, H, W, N = case
    dev, dtype = hip_device, torch.bfloat16
    g = torch.Generator().manual_seed(11 + cin)
    x = torch.randn(N, cin, H, W, generator=g).bfloat16().float()
    geom = ops.ConvGeom(cin, cout, k, s, p, transposed=tr, output_padding=1 if tr else 0)
    Ho, Wo = geom.out_hw(H, W)
    dy = torch.randn(N, cout, Ho, Wo, generator=g).bfloat16().float()
    cin_pad, cout_pad = ops.pad_to(cin, 8), ops.pad_to(cout, 8)
    xd, dyd = nhwc(x, cin_pad, dtype, dev), nhwc(dy, cout_pad, dtype, dev)
    rows, cols = (cin, cout) if tr else (cout, cin)
    dw0 = torch.randn(rows, k * k, cols, generator=g)
    db0 = torch.randn(cout, generator=g)
    d = geom.desc(dtype, N, H, W, cin_pad, cin_pad, cout_pad)
    assert lib().s2p_conv2d_wgrad_workspace(ctypes.byref(d), cin, cout) > 0          # the deterministic path is the one tested

    def run():
        dw = dw0.clone().to(dev); db = None if tr else db0.clone().to(dev)
        ops.conv_wgrad(geom, xd, dyd, dw, cin_pad, cin, cout, db=db)
        torch.cuda.synchronize()
        return dw.cpu(), (None if tr else db.cpu())

    dw, db = run()
    xr = x.double()
    wr = torch.zeros((cin, cout, k, k) if tr else (cout, cin, k, k), dtype=torch.float64, requires_grad=True)
    if tr:
        F.conv_transpose2d(xr, wr, stride=s, padding=p, output_padding=1).backward(dy.double())
    else:
        F.conv2d(xr, wr, stride=s, padding=p).backward(dy.double())
    ref = dw0.double() + wr.grad.permute(0, 2, 3, 1).reshape(rows, k * k, cols)
    assert rel_err(dw, ref) < 1e-5
    if not tr:
        assert rel_err(db, db0.double() + dy.double().sum((0, 2, 3))) < 1e-5
    dw2, db2 = run()
    assert torch.equal(dw, dw2)
    if not tr:
        assert torch.equal(db, db2)


@pytest.mark.parametrize("case", [(512, 4, 13, 13, 5), (128, 4, 8, 7, 3), (64, 3, 9, 9, 2), (512, 4, 13, 13, 128), (512, 4, 8, 8, 128), (512, 4, 3, 5, 2)])
def test_conv_wgrad_head(hip_device, case):
    """PatchGAN logit head (Cout = 1, stride 1, pad 2): activation-stationary wgrad (csrc/wgrad_head.hip) behind
    s2p_conv2d_wgrad_batched, against float64 autograd; accumulates into dw / db; bitwise reproducible.  Cin = 512 with 4x4 taps
    runs the streaming form (the activation through per-wave LDS-DMA rings), incl. the production batch of both scales."""
    cin, k, H, W, N = case
    dev = hip_device
    g = torch.Generator().manual_seed(cin + k)
    pad = 2
    x = torch.randn(N, cin, H, W, generator=g).bfloat16().float()
    geom = ops.ConvGeom(cin, 1, k, 1, pad)
    Ho, Wo = geom.out_hw(H, W)
    dy = torch.randn(N, 1, Ho, Wo, generator=g).bfloat16().float()
    dw0 = torch.randn(1, k * k, cin, generator=g); db0 = torch.randn(1, generator=g)
    xd = nhwc(x, cin, torch.bfloat16, dev); dyd = nhwc(dy, 8, torch.bfloat16, dev)

    def run():
        dw = dw0.clone().to(dev); db = db0.clone().to(dev)
        ops.conv_wgrad_batched(geom, [(xd, 0, dyd, 0, dw, db)], cin, cin, 1)
        torch.cuda.synchronize()
        return dw.cpu(), db.cpu()

    dw, db = run()
    wr = torch.zeros(1, cin, k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wr, padding=pad).backward(dy.double())
    assert rel_err(dw, dw0.double() + wr.grad.permute(0, 2, 3, 1).reshape(1, k * k, cin)) < 1e-5
    assert rel_err(db, db0.double() + dy.double().sum()) < 1e-5
    dw2, db2 = run()
    assert torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.parametrize("case", [(5, 357, 256, True), (64, 256, 6144, False), (130, 40, 72, True), (64, 256, 256, True)])
def test_linear_small_fwd_bwd(hip_device, case):
    """State-path linear layers (csrc/linear_small.hip): y = lrelu(x W^T + b), and the backward with the LeakyReLU
    derivative folded into the operands (dW, db accumulate; dx; the 6144-deep dgrad runs split-K + fixed-order reduce)
    against float64 autograd at 1e-5; bitwise reproducible."""
    M, K, N, lrelu = case
    dev = hip_device
    g = torch.Generator().manual_seed(K + N)
    Kp = ops.pad_to(K, 4)
    x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / math.sqrt(K); b = torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    pre = F.linear(xr, wr, br)
    yr = F.leaky_relu(pre, 0.2) if lrelu else pre
    yr.backward(dy.double())
    xp = torch.zeros(M, Kp); xp[:, :K] = x
    wf = torch.zeros(1, N, 1, Kp); wf[0, :, 0, :K] = w
    Np = ops.pad_to(N, 4)
    wb = torch.zeros(1, Kp, 1, Np); wb[0, :K, 0, :N] = w.t()
    act = ACT_LRELU if lrelu else ACT_NONE
    y = ops.linear_fwd(xp.to(dev), wf.to(dev), b.to(dev), Kp, N, act, 0.2)
    torch.cuda.synchronize()
    assert y.shape == (M, Np) and rel_err(y[:, :N].cpu(), yr.detach()) < 1e-5
    dw0 = torch.randn(N, K, generator=g); db0 = torch.randn(N, generator=g)
    dyp = torch.zeros(M, Np); dyp[:, :N] = dy

    def run():
        dw = dw0.clone().to(dev); db = db0.clone().to(dev)
        dx = ops.linear_bwd(xp.to(dev), dyp.to(dev), y if lrelu else None, wb.to(dev), Kp, K, N, act, 0.2, dw, db)
        torch.cuda.synchronize()
        return dw.cpu(), db.cpu(), dx.cpu()

    dw, db, dx = run()
    assert rel_err(dw, dw0.double() + wr.grad) < 1e-5
    assert rel_err(db, db0.double() + br.grad) < 1e-5
    assert rel_err(dx[:, :K], xr.grad) < 1e-5 and float(dx[:, K:].abs().max() if Kp > K else 0.0) == 0.0
    dw2, db2, dx2 = run()
    assert torch.equal(dw, dw2) and torch.equal(db, db2) and torch.equal(dx, dx2)


@pytest.mark.parametrize("dtype,HW", [(torch.bfloat16, 21), (torch.float32, 9), (torch.bfloat16, 40)])
def test_instance_norm_backward_with_fused_residual(hip_device, dtype, HW):
    """s2p_in_norm_bwd_res: dx + res in one launch must equal the two-launch form (backward, then add) to one rounding, on the
    fused small-plane kernel (21x21, 9x9) and on the two-kernel path (40x40: the residual is a third pass there)."""
    dev = hip_device
    g = torch.Generator().manual_seed(3)
    N, C = 3, 128
    x = (torch.randn(N, HW, HW, C, generator=g) * 1.5 + 0.3).to(dtype).to(dev)
    da = torch.randn(N, HW, HW, C, generator=g).to(dtype).to(dev)
    res = torch.randn(N, HW, HW, C, generator=g).to(dtype).to(dev)
    y, stats = ops.in_norm_fwd(x, C, act=ACT_LRELU, slope=0.2)
    dx0 = ops.in_bwd(da, x, C, stats, act=ACT_LRELU, slope=0.2)
    dx1 = ops.in_bwd(da, x, C, stats, act=ACT_LRELU, slope=0.2, res=res)
    ref = dx0.float() + res.float()
    tol = 1e-6 if dtype == torch.float32 else 1.6e-2          # bf16: dx0 was rounded once before the add, dx1 only after it
    err = float((dx1.float() - ref).abs().max() / ref.abs().max())
    assert err < tol, err


@pytest.mark.parametrize("dtype,shape", [
    (torch.bfloat16, (8, 128, 128, 21, 21)),     # plane-resident kernel: conv + residual + IN + MAT + LeakyReLU in ONE launch
    (torch.bfloat16, (3, 64, 192, 19, 20)),      # ... N % 8 != 0, three co slabs, 380-px plane
    (torch.bfloat16, (2, 64, 64, 9, 7)),         # small plane: conv launch + norm launch behind the same entry point
    (torch.float32, (2, 16, 32, 21, 21)),        # fp32 parity path: two launches
])
@pytest.mark.parametrize("residual", [False, True])
def test_conv_fwd_mat_fused(hip_device, dtype, shape, residual):
    """s2p_conv2d_fwd_mat == F.conv2d (+ skip) -> F.instance_norm -> x_hat * (1 + gamma) + beta -> LeakyReLU, and the
    statistics buffer it leaves is the one s2p_in_norm_bwd consumes."""
    dev = hip_device
    N, cin, cout, H, W = shape
    if residual and cin != cout:
        pytest.skip("residual needs cin == cout")
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    gb = torch.randn(N, 2 * cout + 16, H, W, generator=g) * 0.5      # gamma | beta at a channel offset inside a wider tensor
    st = torch.randn(N, 2 * cout + 8, generator=g) * 0.5
    if dtype == torch.bfloat16:
        x = x.bfloat16().float(); w = w.bfloat16().float(); gb = gb.bfloat16().float()
    geom = ops.ConvGeom(cin, cout, 3, 1, 1)
    xd = nhwc(x, cin, dtype, dev)
    wf = pack_fwd(w, cin, dtype, dev)
    gbd = nhwc(gb, 2 * cout + 16, dtype, dev)
    std = st.to(dev)
    y, ym, stats = ops.conv_fwd_mat(geom, xd, wf, b.to(dev), cin, gbd, 16, std, 8, act=ACT_LRELU, slope=0.2,
                                    aux=xd if residual else None, epi=EPI_ADD if residual else EPI_STORE)
    torch.cuda.synchronize()
    y_ref = F.conv2d(x.double(), w.double(), b.double(), padding=1) + (x.double() if residual else 0)
    assert rel_err(nchw(y, cout), y_ref) < TOL[dtype]
    # the norm is applied to the tensor the kernel STORED (bf16-rounded on the bf16 path)
    ys = nchw(y, cout).double()
    mean = ys.mean((2, 3), keepdim=True); var = ys.var((2, 3), unbiased=False, keepdim=True)
    xh = (ys - mean) / torch.sqrt(var + 1e-5)
    gam = gb[:, 16:16 + cout].double() + st[:, 8:8 + cout].double()[:, :, None, None]
    bet = gb[:, 16 + cout:16 + 2 * cout].double() + st[:, 8 + cout:8 + 2 * cout].double()[:, :, None, None]
    ym_ref = F.leaky_relu(xh * (1 + gam) + bet, 0.2)
    assert rel_err(nchw(ym, cout), ym_ref) < (1e-5 if dtype == torch.float32 else 6e-3)
    # same result as the two separate calls, and the statistics buffer feeds the norm backward
    ym2, stats2 = ops.in_norm_fwd(y, cout, gbd, 16, std, 8, ACT_LRELU, 0.2)
    assert rel_err(ym.float().cpu(), ym2.float().cpu().double()) < (1e-5 if dtype == torch.float32 else 6e-3)
    da = torch.randn(N, cout, H, W, generator=g)
    if dtype == torch.bfloat16:
        da = da.bfloat16().float()
    dad = nhwc(da, cout, dtype, dev)
    dx1 = ops.in_bwd(dad, y, cout, stats, gbd, 16, std, 8, ACT_LRELU, 0.2, torch.empty_like(gbd), 16, torch.empty_like(std), 8)
    dx2 = ops.in_bwd(dad, y, cout, stats2, gbd, 16, std, 8, ACT_LRELU, 0.2, torch.empty_like(gbd), 16, torch.empty_like(std), 8)
    torch.cuda.synchronize()
    assert rel_err(dx1.float().cpu(), dx2.float().cpu().double()) < (1e-5 if dtype == torch.float32 else 6e-3)
    # a forward that keeps nothing for a backward: where the launch is fused the conv output is not written (y is None), the
    # modulated tensor and the statistics are the same bits
    y0, ym0, stats0 = ops.conv_fwd_mat(geom, xd, wf, b.to(dev), cin, gbd, 16, std, 8, act=ACT_LRELU, slope=0.2,
                                       aux=xd if residual else None, epi=EPI_ADD if residual else EPI_STORE, want_y=False)
    nst = 4 + N * cout * 2                                   # header + [N][C]{mean, M2} of a one-split statistics buffer (the rest is never written)
    assert torch.equal(ym0, ym) and torch.equal(stats0[:nst], stats[:nst])
    assert (y0 is None) or torch.equal(y0, y)


@pytest.mark.parametrize("dtype,shape", [
    (torch.bfloat16, (8, 128, 128, 21, 21)),     # plane-resident kernel: dgrad + MAT-norm backward in ONE launch
    (torch.bfloat16, (3, 192, 64, 19, 20)),      # ... N % 8 != 0, 380-px plane, three input-channel slabs produced
    (torch.bfloat16, (2, 64, 64, 9, 7)),         # small plane: two launches behind the same entry point
    (torch.float32, (2, 16, 32, 21, 21)),        # fp32 parity path: two launches
])
@pytest.mark.parametrize("with_res", [False, True])
def test_conv_dgrad_mat_fused(hip_device, dtype, shape, with_res):
    """s2p_conv2d_dgrad_mat == conv dgrad -> backward of (InstanceNorm -> MAT modulation -> LeakyReLU) [+ skip gradient],
    against the float64 autograd of the same chain and against the two separate HIP calls."""
    dev = hip_device
    N, C, cout, H, W = shape                      # the norm has C channels and feeds a conv C -> cout
    g = torch.Generator().manual_seed(13)
    xn = torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3
    w = torch.randn(cout, C, 3, 3, generator=g) / math.sqrt(C * 9)
    gb = torch.randn(N, 2 * C + 16, H, W, generator=g) * 0.5
    st = torch.randn(N, 2 * C + 8, generator=g) * 0.5
    dy = torch.randn(N, cout, H, W, generator=g)
    res = torch.randn(N, C, H, W, generator=g)
    if dtype == torch.bfloat16:
        xn = xn.bfloat16().float(); w = w.bfloat16().float(); gb = gb.bfloat16().float(); dy = dy.bfloat16().float(); res = res.bfloat16().float()
    geom = ops.ConvGeom(C, cout, 3, 1, 1)
    xnd = nhwc(xn, C, dtype, dev)
    wb = pack_bwd(w, C, cout, dtype, dev)
    gbd = nhwc(gb, 2 * C + 16, dtype, dev)
    std = st.to(dev)
    dyd = nhwc(dy, cout, dtype, dev)
    resd = nhwc(res, C, dtype, dev) if with_res else None
    a_fwd, stats = ops.in_norm_fwd(xnd, C, gbd, 16, std, 8, ACT_LRELU, 0.2)
    dgb1 = torch.zeros_like(gbd); dst1 = torch.zeros_like(std)
    dx1 = ops.conv_dgrad_mat(geom, dyd, wb, xnd, C, stats, gbd, 16, std, 8, ACT_LRELU, 0.2, dgb1, 16, dst1, 8, res=resd)
    # the two separate calls
    dgb2 = torch.zeros_like(gbd); dst2 = torch.zeros_like(std)
    d_mid = ops.conv_dgrad(geom, dyd, wb, tuple(xnd.shape), C)
    dx2 = ops.in_bwd(d_mid, xnd, C, stats, gbd, 16, std, 8, ACT_LRELU, 0.2, dgb2, 16, dst2, 8, res=resd)
    torch.cuda.synchronize()
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(dx1.float().cpu(), dx2.float().cpu().double()) < tol
    assert rel_err(dgb1.float().cpu(), dgb2.float().cpu().double()) < tol
    assert rel_err(dst1.cpu(), dst2.cpu().double()) < (1e-5 if dtype == torch.float32 else 2e-3)
    # float64 autograd of norm -> modulate -> lrelu -> conv
    xr = xn.double().requires_grad_(True); gbr = gb.double().requires_grad_(True); str_ = st.double().requires_grad_(True)
    xh = F.instance_norm(xr, eps=1e-5)
    gam = gbr[:, 16:16 + C] + str_[:, 8:8 + C][:, :, None, None]
    bet = gbr[:, 16 + C:16 + 2 * C] + str_[:, 8 + C:8 + 2 * C][:, :, None, None]
    y = F.conv2d(F.leaky_relu(xh * (1 + gam) + bet, 0.2), w.double(), padding=1)
    y.backward(dy.double())
    ref_dx = xr.grad + (res.double() if with_res else 0)
    assert rel_err(nchw(dx1, C), ref_dx) < TOL[dtype]
    assert rel_err(nchw(dgb1, 2 * C + 16)[:, 16:16 + 2 * C], gbr.grad[:, 16:16 + 2 * C]) < TOL[dtype]
    assert rel_err(dst1.cpu()[:, 8:8 + 2 * C], str_.grad[:, 8:8 + 2 * C]) < (1e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("N,HW_", [(2, (21, 21)), (16, (21, 21)), (5, (18, 20)), (130, (16, 21))])      # the last: 260 / 780 tiles -> PAIR kernel
def test_conv_plane_kernels_epilogues_and_groups(hip_device, N, HW_):
    """Every epilogue form of the plane-resident kernels (conv_plane.hip) on shapes that take them -- N * Cout / 64 <= 256 tiles:
    the K-split kernel; more: the PAIR kernel -- against float64: bias + ReLU, residual add, the dgrad with the fused
    producer-activation gradient and a second incoming gradient (the VGG taps), and a grouped conv with bias (the gamma/beta heads)."""
    dev = hip_device
    dtype = torch.bfloat16
    H, W = HW_
    C = 128
    g = torch.Generator().manual_seed(3)
    r = lambda *sh: torch.randn(*sh, generator=g).bfloat16().float()      # noqa: E731
    x, w, b = r(N, C, H, W), r(C, C, 3, 3) / math.sqrt(C * 9), torch.randn(C, generator=g)
    geom = ops.ConvGeom(C, C, 3, 1, 1)
    xd, wf, wb = nhwc(x, C, dtype, dev), pack_fwd(w, C, dtype, dev), pack_bwd(w, C, C, dtype, dev)
    y = ops.conv_fwd(geom, xd, wf, b.to(dev), C, act=ACT_RELU)
    assert rel_err(nchw(y, C), F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))) < TOL[dtype]
    y = ops.conv_fwd(geom, xd, wf, None, C, aux=xd, epi=EPI_ADD)
    assert rel_err(nchw(y, C), x.double() + F.conv2d(x.double(), w.double(), padding=1)) < TOL[dtype]
    a, dy, d2 = F.relu(x), r(N, C, H, W), r(N, C, H, W)
    dx = ops.conv_dgrad(geom, nhwc(dy, C, dtype, dev), wb, tuple(xd.shape), C, aux=nhwc(a, C, dtype, dev), epi=EPI_MUL_ACTGRAD,
                        aux_act=ACT_RELU, aux2=nhwc(d2, C, dtype, dev))
    ref = (F.conv_transpose2d(dy.double(), w.double(), padding=1) + d2.double()) * (a > 0).double()
    assert rel_err(nchw(dx, C), ref) < TOL[dtype]
    # grouped: 3 groups of 64 -> 128 with bias, LeakyReLU
    G, ci, co = 3, 64, 128
    xg, wg, bg = r(N, G * ci, H, W), r(G * co, ci, 3, 3) / math.sqrt(ci * 9), torch.randn(G * co, generator=g)
    gg = ops.ConvGeom(ci, co, 3, 1, 1, groups=G, x_gstride=ci, y_gstride=co)
    wfg = torch.stack([pack_fwd(wg[i * co:(i + 1) * co], ci, dtype, dev) for i in range(G)])
    yg = ops.conv_fwd(gg, nhwc(xg, G * ci, dtype, dev), wfg, bg.to(dev), ci, y_pitch=G * co, act=ACT_LRELU, slope=0.2)
    assert rel_err(nchw(yg, G * co), F.leaky_relu(F.conv2d(xg.double(), wg.double(), bg.double(), padding=1, groups=G), 0.2)) < TOL[dtype]
    wbg = torch.stack([pack_bwd(wg[i * co:(i + 1) * co], ci, co, dtype, dev) for i in range(G)])
    dyg = r(N, G * co, H, W)
    dxg = ops.conv_dgrad(gg, nhwc(dyg, G * co, dtype, dev), wbg, (N, H, W, G * ci), ci)
    refg = torch.cat([F.conv_transpose2d(dyg[:, i * co:(i + 1) * co].double(), wg[i * co:(i + 1) * co].double(), padding=1) for i in range(G)], 1)
    assert rel_err(nchw(dxg, G * ci), refg) < TOL[dtype]
    # the same grouped conv with its produced side GROUP-MAJOR ([G][N][H][W][co]: group stride >= pitch, the layout of the
    # generator's gamma|beta planes): forward writes it, dgrad and the batched weight gradient read it -- same bits
    ggm = ops.ConvGeom(ci, co, 3, 1, 1, groups=G, x_gstride=ci, y_gstride=N * H * W * co)
    y5 = torch.full((G, N, H, W, co), float("nan"), dtype=dtype, device=dev)
    ops.conv_fwd(ggm, nhwc(xg, G * ci, dtype, dev), wfg, bg.to(dev), ci, y_pitch=co, act=ACT_LRELU, slope=0.2, out=y5[0])
    for i in range(G):
        assert torch.equal(y5[i], yg[..., i * co:(i + 1) * co])
    dy_il = nhwc(dyg, G * co, dtype, dev)
    dy5 = torch.stack([dy_il[..., i * co:(i + 1) * co] for i in range(G)]).contiguous()
    dx5 = ops.conv_dgrad(ggm, dy5[0], wbg, (N, H, W, G * ci), ci)
    assert torch.equal(dx5, dxg)
    one = ops.ConvGeom(ci, co, 3, 1, 1)
    xil = nhwc(xg, G * ci, dtype, dev)
    dw_a = torch.zeros(G, co * 9 * ci, device=dev); db_a = torch.zeros(G, co, device=dev)
    dw_b = torch.zeros_like(dw_a); db_b = torch.zeros_like(db_a)
    ops.conv_wgrad_batched(one, [(xil, i * ci, dy_il, i * co, dw_a[i], db_a[i]) for i in range(G)], ci, ci, co)
    ops.conv_wgrad_batched(one, [(xil, i * ci, dy5[i], 0, dw_b[i], db_b[i]) for i in range(G)], ci, ci, co)
    torch.cuda.synchronize()
    assert torch.equal(dw_a, dw_b) and torch.equal(db_a, db_b)
    refw = torch.cat([torch.autograd.grad(F.conv2d(xg[:, i * ci:(i + 1) * ci].double(), wv, padding=1), wv, dyg[:, i * co:(i + 1) * co].double())[0]
                      for i in range(G) for wv in [wg[i * co:(i + 1) * co].double().requires_grad_(True)]])
    assert rel_err(dw_b.view(G * co, 3, 3, ci).permute(0, 3, 1, 2).cpu(), refw) < TOL[dtype]


@pytest.mark.parametrize("shape", [
    (8, 256, 512, 12, 12),      # PatchGAN 256 -> 512 4x4 stride 1 pad 2 on the finer scale's 12x12 map: 169-px planes, 3 pixel blocks per wave
    (3, 64, 128, 7, 7),         # ... the coarser scale's 7x7 map: 64-px planes, one pixel block per wave; N % 8 != 0
    (5, 128, 64, 11, 10),       # ragged plane 12 x 11, one output slab
    (64, 256, 512, 12, 12),     # the production batch (XCD-remap branch), references on 4 samples
])
def test_conv_planeg_4x4_stride1(hip_device, shape):
    """The generalised plane-resident kernel (csrc/conv_planeg.hip) on the PatchGAN 4x4 stride-1 layers, against float64:
    forward (plain, and conv -> InstanceNorm -> LeakyReLU in ONE launch through s2p_conv2d_fwd_mat), dgrad (plain, and
    dgrad (+ tap gradient) -> InstanceNorm / LeakyReLU backward in ONE launch through s2p_conv2d_dgrad_mat)."""
    import ctypes
    from s2p_amd import _lib
    dev = hip_device
    dtype = torch.bfloat16
    N, cin, cout, H, W = shape
    pick = list(range(N)) if N <= 8 else [0, 21, 42, 63]
    g = torch.Generator().manual_seed(23)
    r = lambda *sh: torch.randn(*sh, generator=g).bfloat16().float()      # noqa: E731
    x, w = r(N, cin, H, W), r(cout, cin, 4, 4) / math.sqrt(cin * 16)
    geom = ops.ConvGeom(cin, cout, 4, 1, 2)
    Ho, Wo = geom.out_hw(H, W)
    d = geom.desc(dtype, N, H, W, cin, cin, cout)
    assert _lib.lib().s2p_conv2d_mat_is_fused(ctypes.byref(d), 0, 0) == 1 and _lib.lib().s2p_conv2d_mat_is_fused(ctypes.byref(d), 1, 0) == 1
    xd, wf, wb = nhwc(x, cin, dtype, dev), pack_fwd(w, cin, dtype, dev), pack_bwd(w, cin, cout, dtype, dev)
    # forward
    y = ops.conv_fwd(geom, xd, wf, None, cin, act=ACT_LRELU, slope=0.2)
    y_ref = F.conv2d(x[pick].double(), w.double(), padding=2)
    assert rel_err(nchw(y, cout)[pick], F.leaky_relu(y_ref, 0.2)) < TOL[dtype]
    c, f, stats = ops.conv_fwd_mat(geom, xd, wf, None, cin, None, 0, None, 0, act=ACT_LRELU, slope=0.2)
    torch.cuda.synchronize()
    assert rel_err(nchw(c, cout)[pick], y_ref) < TOL[dtype]
    cs = nchw(c, cout)[pick].double()                           # the norm acts on the tensor the kernel STORED
    assert rel_err(nchw(f, cout)[pick], F.leaky_relu(F.instance_norm(cs, eps=1e-5), 0.2)) < 6e-3
    f2, stats2 = ops.in_norm_fwd(c, cout, act=ACT_LRELU, slope=0.2)      # the separate norm launch: same tensor, same statistics
    assert rel_err(f.float().cpu(), f2.float().cpu().double()) < 6e-3
    hdr = stats[:2].view(torch.int32).cpu().tolist()           # norm.hip's self-describing format: {splits, rows per split}
    assert hdr == [1, Ho * Wo]
    mom = stats[4:4 + N * cout * 2].view(N, cout, 2).cpu()[pick].double()
    assert rel_err(mom[..., 0], cs.mean((2, 3))) < 1e-4 and rel_err(mom[..., 1], cs.var((2, 3), unbiased=False) * (Ho * Wo)) < 1e-3
    # dgrad: dy on the Ho x Wo grid -> dx on H x W
    dy = r(N, cout, Ho, Wo)
    dyd = nhwc(dy, cout, dtype, dev)
    dx = ops.conv_dgrad(geom, dyd, wb, tuple(xd.shape), cin)
    assert rel_err(nchw(dx, cin)[pick], F.conv_transpose2d(dy[pick].double(), w.double(), padding=2)) < TOL[dtype]
    # dgrad (+ tap gradient on the activation) -> backward of InstanceNorm + LeakyReLU of the layer in front (its output = x here)
    xn, tap = (r(N, cin, H, W) * 1.5 + 0.3).bfloat16().float(), r(N, cin, H, W)      # (bf16-exact: the reference sees the same branches)
    xnd, tapd = nhwc(xn, cin, dtype, dev), nhwc(tap, cin, dtype, dev)
    _, st_n = ops.in_norm_fwd(xnd, cin, act=ACT_LRELU, slope=0.2)
    for with_tap in (False, True):
        dxn = ops.conv_dgrad_mat(geom, dyd, wb, xnd, cin, st_n, None, 0, None, 0, ACT_LRELU, 0.2, None, 0, None, 0,
                                 aux=tapd if with_tap else None)
        torch.cuda.synchronize()
        xr = xn[pick].double().requires_grad_(True)
        a = F.leaky_relu(F.instance_norm(xr, eps=1e-5), 0.2)
        loss = (F.conv2d(a, w.double(), padding=2) * dy[pick].double()).sum() + ((a * tap[pick].double()).sum() if with_tap else 0)
        loss.backward()
        assert rel_err(nchw(dxn, cin)[pick], xr.grad) < TOL[dtype], with_tap
        # and the two separate launches
        d_mid = ops.conv_dgrad(geom, dyd, wb, tuple(xd.shape), cin, aux=tapd if with_tap else None, epi=EPI_ADD if with_tap else EPI_STORE)
        dx2 = ops.in_bwd(d_mid, xnd, cin, st_n, act=ACT_LRELU, slope=0.2)
        assert rel_err(dxn.float().cpu(), dx2.float().cpu().double()) < 1e-2


@pytest.mark.parametrize("shape", [(8, 512, 512, 10, 10), (3, 256, 512, 10, 10), (5, 64, 128, 9, 11), (64, 512, 512, 10, 10),
                                   (64, 512, 512, 5, 5),      # VGG conv5_1 at the production batch: FOUR images stacked in one plane
                                   (32, 64, 128, 6, 4),       # two images per plane (N / 4 < 16 planes), ragged 6 x 4
                                   (64, 128, 64, 3, 7)])      # four images of 21 pixels, wider than tall
def test_conv_planeg_3x3_small_planes(hip_device, shape):
    """3x3 stride-1 pad-1 convs on planes of 65..128 pixels (VGG conv4_x at 10x10; 9x11 ragged) on the generalised plane-resident
    kernel (csrc/conv_planeg.hip): forward with bias + ReLU, and the dgrad with the fused producer-activation gradient and a second
    incoming gradient (the VGG taps), against float64.  Planes of fewer than 64 pixels (round 5): 2 or 4 consecutive images share a
    workgroup, stacked in one padded raster with a common zero row between them -- every image of the batch is compared."""
    dev = hip_device
    dtype = torch.bfloat16
    N, cin, cout, H, W = shape
    pick = list(range(N)) if (N <= 8 or H * W < 64) else [0, 21, 42, 63]
    g = torch.Generator().manual_seed(31)
    r = lambda *sh: torch.randn(*sh, generator=g).bfloat16().float()      # noqa: E731
    x, w, b = r(N, cin, H, W), r(cout, cin, 3, 3) / math.sqrt(cin * 9), torch.randn(cout, generator=g)
    geom = ops.ConvGeom(cin, cout, 3, 1, 1)
    xd, wf, wb = nhwc(x, cin, dtype, dev), pack_fwd(w, cin, dtype, dev), pack_bwd(w, cin, cout, dtype, dev)
    y = ops.conv_fwd(geom, xd, wf, b.to(dev), cin, act=ACT_RELU)
    assert rel_err(nchw(y, cout)[pick], F.relu(F.conv2d(x[pick].double(), w.double(), b.double(), padding=1))) < TOL[dtype]
    a, dy, d2 = F.relu(x), r(N, cout, H, W), r(N, cin, H, W)
    dx = ops.conv_dgrad(geom, nhwc(dy, cout, dtype, dev), wb, tuple(xd.shape), cin, aux=nhwc(a, cin, dtype, dev), epi=EPI_MUL_ACTGRAD,
                        aux_act=ACT_RELU, aux2=nhwc(d2, cin, dtype, dev))
    ref = (F.conv_transpose2d(dy[pick].double(), w.double(), padding=1) + d2[pick].double()) * (a[pick] > 0).double()
    assert rel_err(nchw(dx, cin)[pick], ref) < TOL[dtype]


@pytest.mark.parametrize("shape", [
    (8, 128, 256, 22, 22),      # PatchGAN 128 -> 256 4x4 stride 2 pad 2, finer scale: 22x22 -> 12x12 (four 11x11 parity sub-planes)
    (3, 64, 128, 22, 22),       # 64 -> 128 of the coarser scale (two 32-channel groups per parity), N % 8 != 0
    (5, 128, 256, 12, 12),      # 128 -> 256 of the coarser scale: 12x12 -> 7x7, one pixel block per wave
    (2, 64, 64, 21, 23),        # odd sizes: the parity sub-planes differ in size (11 / 10 rows, 12 / 11 columns)
    (64, 128, 256, 22, 22),     # the production batch, references on 4 samples
])
def test_conv_planeg_4x4_stride2(hip_device, shape):
    """The generalised plane-resident kernel in its PARITY form (csrc/conv_planeg.hip): a 4x4 stride-2 pad-2 convolution as a 2x2
    stride-1 convolution over the four parity sub-planes of the input, the space-to-depth done by the LDS-DMA source addresses.
    Forward, plain and with InstanceNorm + LeakyReLU in the epilogue (s2p_conv2d_fwd_mat), against float64."""
    import ctypes
    from s2p_amd import _lib
    dev = hip_device
    dtype = torch.bfloat16
    N, cin, cout, H, W = shape
    pick = list(range(N)) if N <= 8 else [0, 21, 42, 63]
    g = torch.Generator().manual_seed(29)
    r = lambda *sh: torch.randn(*sh, generator=g).bfloat16().float()      # noqa: E731
    x, w, b = r(N, cin, H, W), r(cout, cin, 4, 4) / math.sqrt(cin * 16), torch.randn(cout, generator=g)
    geom = ops.ConvGeom(cin, cout, 4, 2, 2)
    Ho, Wo = geom.out_hw(H, W)
    d = geom.desc(dtype, N, H, W, cin, cin, cout)
    assert _lib.lib().s2p_conv2d_mat_is_fused(ctypes.byref(d), 0, 0) == 1
    xd, wf = nhwc(x, cin, dtype, dev), pack_fwd(w, cin, dtype, dev)
    y = ops.conv_fwd(geom, xd, wf, b.to(dev), cin, act=ACT_LRELU, slope=0.2)
    y_ref = F.conv2d(x[pick].double(), w.double(), b.double(), stride=2, padding=2)
    assert tuple(y.shape[1:3]) == (Ho, Wo) == tuple(y_ref.shape[2:])
    assert rel_err(nchw(y, cout)[pick], F.leaky_relu(y_ref, 0.2)) < TOL[dtype]
    c, f, stats = ops.conv_fwd_mat(geom, xd, wf, None, cin, None, 0, None, 0, act=ACT_LRELU, slope=0.2)
    torch.cuda.synchronize()
    assert rel_err(nchw(c, cout)[pick], y_ref - b.double()[None, :, None, None]) < TOL[dtype]
    cs = nchw(c, cout)[pick].double()
    assert rel_err(nchw(f, cout)[pick], F.leaky_relu(F.instance_norm(cs, eps=1e-5), 0.2)) < 6e-3
    mom = stats[4:4 + N * cout * 2].view(N, cout, 2).cpu()[pick].double()
    assert rel_err(mom[..., 0], cs.mean((2, 3))) < 1e-4


@pytest.mark.parametrize("residual", [False, True])
def test_plane_kernels_production_shape(hip_device, residual):
    """The production shapes of the timed workload (missing #2), at full size, against float64 on samples
    {0, 21, 42, 63} (every op is per-sample): (64, 256, 256, 21, 21) bf16 through s2p_conv2d_fwd_mat -- the fused
    `conv_plane_kernel<7,22,0,1>` on its (N & 7) == 0 XCD-remap branch, one workgroup per CU -- and through
    s2p_conv2d_dgrad_mat (`<7,22,0,2>`)."""
    dev = hip_device
    dtype = torch.bfloat16
    N, C, H, W = 64, 256, 21, 21
    pick = [0, 21, 42, 63]
    g = torch.Generator().manual_seed(17)
    r = lambda *sh: torch.randn(*sh, generator=g).bfloat16().float()      # noqa: E731
    x, w, b = r(N, C, H, W), r(C, C, 3, 3) / math.sqrt(C * 9), torch.randn(C, generator=g)
    gb = r(N, 2 * C, H, W) * 0.5
    st = torch.randn(N, 2 * C, generator=g) * 0.5
    geom = ops.ConvGeom(C, C, 3, 1, 1)
    xd, wf, wb = nhwc(x, C, dtype, dev), pack_fwd(w, C, dtype, dev), pack_bwd(w, C, C, dtype, dev)
    gbd, std = nhwc(gb, 2 * C, dtype, dev), st.to(dev)
    y, ym, stats = ops.conv_fwd_mat(geom, xd, wf, b.to(dev), C, gbd, 0, std, 0, act=ACT_LRELU, slope=0.2,
                                    aux=xd if residual else None, epi=EPI_ADD if residual else EPI_STORE)
    torch.cuda.synchronize()
    xs = x[pick].double()
    y_ref = F.conv2d(xs, w.double(), b.double(), padding=1) + (xs if residual else 0)
    assert rel_err(nchw(y, C)[pick], y_ref) < TOL[dtype]
    ys = nchw(y, C)[pick].double()                               # the norm acts on the tensor the kernel STORED
    xh = F.instance_norm(ys, eps=1e-5)
    gam = gb[pick, :C].double() + st[pick, :C].double()[:, :, None, None]
    bet = gb[pick, C:].double() + st[pick, C:].double()[:, :, None, None]
    assert rel_err(nchw(ym, C)[pick], F.leaky_relu(xh * (1 + gam) + bet, 0.2)) < 6e-3
    # backward form: dgrad of a conv fed by norm(xn) -> norm backward (+ skip gradient)
    dy, res = r(N, C, H, W), r(N, C, H, W)
    xnd, dyd = xd, nhwc(dy, C, dtype, dev)
    _, stats_n = ops.in_norm_fwd(xnd, C, gbd, 0, std, 0, ACT_LRELU, 0.2)
    dgb = torch.zeros_like(gbd); dst = torch.zeros_like(std)
    dx = ops.conv_dgrad_mat(geom, dyd, wb, xnd, C, stats_n, gbd, 0, std, 0, ACT_LRELU, 0.2, dgb, 0, dst, 0,
                            res=nhwc(res, C, dtype, dev) if residual else None)
    torch.cuda.synchronize()
    xr = x[pick].double().requires_grad_(True); gbr = gb[pick].double().requires_grad_(True); str_ = st[pick].double().requires_grad_(True)
    gam = gbr[:, :C] + str_[:, :C][:, :, None, None]
    bet = gbr[:, C:] + str_[:, C:][:, :, None, None]
    F.conv2d(F.leaky_relu(F.instance_norm(xr, eps=1e-5) * (1 + gam) + bet, 0.2), w.double(), padding=1).backward(dy[pick].double())
    assert rel_err(nchw(dx, C)[pick], xr.grad + (res[pick].double() if residual else 0)) < TOL[dtype]
    assert rel_err(nchw(dgb, 2 * C)[pick], gbr.grad) < TOL[dtype]
    assert rel_err(dst.cpu()[pick], str_.grad) < 1e-2


def test_plane_pair_kernel_production_groups(hip_device):
    """The generator's gamma|beta conv at full size: 12 groups of 128 -> 512 at N = 64, 21x21, group-major output (24 tiles per
    CU: `conv_plane_pair_kernel`), forward and dgrad against float64 on samples {0, 21, 42, 63} of groups {0, 5, 11}."""
    dev = hip_device
    dtype = torch.bfloat16
    N, H, W, G, ci, co = 64, 21, 21, 12, 128, 512
    pick, gpick = [0, 21, 42, 63], [0, 5, 11]
    g = torch.Generator().manual_seed(19)
    r = lambda *sh: torch.randn(*sh, generator=g).bfloat16().float()      # noqa: E731
    xg, wg, bg = r(N, G * ci, H, W), r(G * co, ci, 3, 3) / math.sqrt(ci * 9), torch.randn(G * co, generator=g)
    ggm = ops.ConvGeom(ci, co, 3, 1, 1, groups=G, x_gstride=ci, y_gstride=N * H * W * co)
    wfg = torch.stack([pack_fwd(wg[i * co:(i + 1) * co], ci, dtype, dev) for i in range(G)])
    wbg = torch.stack([pack_bwd(wg[i * co:(i + 1) * co], ci, co, dtype, dev) for i in range(G)])
    xd = nhwc(xg, G * ci, dtype, dev)
    y5 = torch.full((G, N, H, W, co), float("nan"), dtype=dtype, device=dev)
    ops.conv_fwd(ggm, xd, wfg, bg.to(dev), ci, y_pitch=co, act=ACT_NONE, out=y5[0])
    dy5 = torch.randn(G, N, H, W, co, generator=g).to(dtype).to(dev)
    dx = ops.conv_dgrad(ggm, dy5[0], wbg, (N, H, W, G * ci), ci)
    # ... and with the producer-activation-gradient epilogue the generator's backward uses (aux = the ReLU'd conditioning features):
    # the persistent form applies it from registers, 8 bytes per lane
    dx2 = ops.conv_dgrad(ggm, dy5[0], wbg, (N, H, W, G * ci), ci, aux=xd, epi=EPI_MUL_ACTGRAD, aux_act=ACT_RELU)
    torch.cuda.synchronize()
    assert not bool(torch.isnan(y5.float()).any())
    for i in gpick:
        wi = wg[i * co:(i + 1) * co].double()
        ref = F.conv2d(xg[pick, i * ci:(i + 1) * ci].double(), wi, bg[i * co:(i + 1) * co].double(), padding=1)
        assert rel_err(nchw(y5[i], co)[pick], ref) < TOL[dtype], i
        dyi = dy5[i].float().cpu().permute(0, 3, 1, 2)[pick].double()
        refd = F.conv_transpose2d(dyi, wi, padding=1)
        assert rel_err(nchw(dx, G * ci)[pick, i * ci:(i + 1) * ci], refd) < TOL[dtype], i
        xi = xg[pick, i * ci:(i + 1) * ci].double()
        assert rel_err(nchw(dx2, G * ci)[pick, i * ci:(i + 1) * ci], refd * (xi > 0)) < TOL[dtype], i


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_pack_weights_against_torch(hip_device, dtype):
    """s2p_pack_weights (csrc/misc.hip): fp32 channels-last master [R][T][C] -> the forward operand [R][T][Cpad] (zero pad) and the
    transposed backward operand [C][T][Rrow] at column offset r_off, in the compute dtype: bitwise against torch's cast, over
    ragged shapes (C not a multiple of 4, R not a multiple of the tile, one-tap linear layers, a fused two-member matrix) in ONE
    launch; elements the jobs do not own keep their previous contents."""
    import ctypes
    from s2p_amd import _lib
    dev = hip_device
    g = torch.Generator().manual_seed(5)
    ce = _lib.chunk_elems(dtype)
    shapes = [(64, 9, 64, 0, 64), (100, 9, 3, 0, 100), (70, 16, 6, 0, 70), (256, 1, 357, 0, 256), (33, 49, 17, 0, 33),
              (128, 9, 128, 0, 256), (128, 9, 128, 128, 256), (1, 16, 512, 0, 1), (130, 4, 260, 2, 136)]     # R, T, C, r_off, Rrow(min)
    jobs, keep = [], []
    for (R, T, C, r_off, rrow) in shapes:
        Cpad, Rrow = ops.pad_to(C, ce), ops.pad_to(max(rrow, r_off + R), ce)
        src = torch.randn(R, T, C, generator=g).to(dev)
        fwd = torch.full((R, T, Cpad), 7.0, dtype=dtype, device=dev)
        bwd = torch.full((C, T, Rrow), 7.0, dtype=dtype, device=dev)
        jobs.append(_lib.PackJob(src.data_ptr(), fwd.data_ptr(), bwd.data_ptr(), R, T, C, Cpad, Rrow, r_off, _lib.dtype_id(dtype)))
        keep.append((src, fwd, bwd, R, T, C, Cpad, Rrow, r_off))
    # a forward-only job (need_bwd=False packs)
    srcf = torch.randn(48, 9, 8, generator=g).to(dev); fwdf = torch.full((48, 9, 8), 7.0, dtype=dtype, device=dev)
    jobs.append(_lib.PackJob(srcf.data_ptr(), fwdf.data_ptr(), 0, 48, 9, 8, 8, 48, 0, _lib.dtype_id(dtype)))
    arr = (_lib.PackJob * len(jobs))(*jobs)
    jd = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    mx = max(s.numel() for s, *_ in keep)
    for _ in range(2):
        _lib.check(_lib.lib().s2p_pack_weights(_lib.ptr(jd), len(jobs), mx, _lib.stream()), "s2p_pack_weights")
    torch.cuda.synchronize()
    for (src, fwd, bwd, R, T, C, Cpad, Rrow, r_off) in keep:
        ref_f = torch.zeros(R, T, Cpad, dtype=dtype, device=dev); ref_f[:, :, :C] = src.to(dtype)
        assert torch.equal(fwd, ref_f), (R, T, C)
        ref_b = torch.full((C, T, Rrow), 7.0, dtype=dtype, device=dev); ref_b[:, :, r_off:r_off + R] = src.permute(2, 1, 0).to(dtype)
        assert torch.equal(bwd, ref_b), (R, T, C, r_off)
    assert torch.equal(fwdf, srcf.to(dtype))
