"""Raw (non-autograd) wrappers over the C ABI.  Tensors are explicit NHWC: shape [N, H, W, pitch] (contiguous),
where `pitch` >= channel count and is a multiple of the 16-byte chunk (4 fp32 / 8 bf16 elements).

Each wrapper cites the torch op of the SPADE-lineage model it replaces (the reference checkout itself ships no
generator code: SURVEY.md section 0; lineage README.md:72-75).
"""
import ctypes

import torch

from . import _lib
from ._lib import (ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, EPI_ADD, EPI_MUL_ACTGRAD, EPI_STORE, ConvDesc,
                   check, chunk_elems, dtype_id, lib, ptr, stream)

IN_EPS = 1e-5

# When True, every "side stream" helper of the networks returns the CURRENT stream: the step runs as one serial chain of
# launches.  bench.py sets it for its instrumented roofline step only, so that a kernel's HIP-event duration is that of the
# kernel alone (in the product configuration independent chains overlap on several streams and would inflate each
# other's per-launch times).
SERIALIZE = False


# Optional in-situ profiler used by bench.py's roofline leg: when PROFILE is a list, every MFMA conv launch is
# bracketed by two events on the launch stream and logged with its algorithmic FLOPs (2*MACs, un-padded channels).
PROFILE = None


class _Prof:
    def __init__(self, kind, geom, N, H, W, dtype):
        self.on = PROFILE is not None
        if self.on:
            Ho, Wo = geom.out_hw(H, W)
            pix = N * H * W if geom.transposed else N * Ho * Wo
            self.rec = dict(kind=kind, flops=2.0 * pix * geom.cin * geom.cout * geom.k * geom.k * geom.groups,
                            shape=(N, H, W, geom.cin, geom.cout, geom.k, geom.stride, geom.groups, int(geom.transposed)),
                            dtype=str(dtype), net=getattr(geom, "net", ""))
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def done(self):
        if self.on:
            self.e1.record()
            self.rec["events"] = (self.e0, self.e1)
            PROFILE.append(self.rec)


class _ProfNorm:
    """Same hook for the IN / MAT kernels: algorithmic HBM bytes instead of FLOPs."""

    def __init__(self, kind, x, C, nbytes, modulated):
        self.on = PROFILE is not None
        if self.on:
            N, H, W, _ = x.shape
            self.rec = dict(kind=kind, flops=0.0, bytes=float(nbytes), shape=(N, H, W, C, int(modulated)), dtype=str(x.dtype))
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def done(self):
        if self.on:
            self.e1.record()
            self.rec["events"] = (self.e0, self.e1)
            PROFILE.append(self.rec)


def pad_to(c, ce):
    return (c + ce - 1) // ce * ce


_MAT_FUSED = {}           # (geometry, problem) -> s2p_conv2d_mat_is_fused, asked once (conv_fwd_mat)
SKIP_UNUSED_Y = True      # conv_fwd_mat(want_y=False) really skips the conv output (tools/ab_step.py flips it for same-box A/Bs)


class ConvGeom:
    """Static geometry of one conv layer (F.conv2d / F.conv_transpose2d arguments)."""

    def __init__(self, cin, cout, k, stride=1, pad=0, transposed=False, reflect=False, groups=1,
                 output_padding=0, x_gstride=0, y_gstride=0, net=""):
        self.net = net              # which network the layer belongs to ("G" / "D" / "VGG": bench.py's per-network aggregates)
        self.cin, self.cout, self.k = cin, cout, k
        self.stride, self.pad = stride, pad
        self.transposed, self.reflect, self.groups = transposed, reflect, groups
        self.output_padding = output_padding
        self.x_gstride, self.y_gstride = x_gstride, y_gstride

    def out_hw(self, H, W):
        k, s, p = self.k, self.stride, self.pad
        if self.transposed:
            return ((H - 1) * s - 2 * p + k + self.output_padding, (W - 1) * s - 2 * p + k + self.output_padding)
        return ((H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1)

    def desc(self, dtype, N, H, W, cin_pad, x_pitch, y_pitch):
        Ho, Wo = self.out_hw(H, W)
        return ConvDesc(dtype_id(dtype), N, H, W, cin_pad, x_pitch, Ho, Wo, self.cout, y_pitch,
                        self.k, self.k, self.stride, self.pad, int(self.transposed), int(self.reflect),
                        self.groups, self.x_gstride, self.y_gstride, self.cin if self.groups == 1 else 0)


def conv_fwd(geom, x, w_fwd, bias, cin_pad, y_pitch=None, act=ACT_NONE, slope=0.2, aux=None, epi=EPI_STORE,
             out=None, x_off=0):
    """y = act(conv(x, w) + b) [+ aux]   (F.conv2d / F.conv_transpose2d + bias + activation).
    x_off: first channel of x the conv gathers (a range of the groups of a grouped conv over a wide tensor)."""
    N, H, W, xp = x.shape
    ce = chunk_elems(x.dtype)
    Ho, Wo = geom.out_hw(H, W)
    if y_pitch is None:
        y_pitch = pad_to(geom.cout * geom.groups if geom.groups > 1 else geom.cout, ce)
    y = out if out is not None else torch.empty((N, Ho, Wo, y_pitch), dtype=x.dtype, device=x.device)
    d = geom.desc(x.dtype, N, H, W, cin_pad, xp, y_pitch)
    need = lib().s2p_conv2d_fwd_workspace(ctypes.byref(d), epi)        # > 0: small-map launch that splits K over the idle CUs
    ws = torch.empty(need, dtype=torch.uint8, device=x.device) if need else None
    pr = _Prof("fwd", geom, N, H, W, x.dtype)
    check(lib().s2p_conv2d_fwd_ws(ctypes.byref(d), ptr(x) + x_off * x.element_size(), ptr(w_fwd), ptr(bias), ptr(aux), ptr(y), act, slope, epi,
                                  ptr(ws), need, stream()), "s2p_conv2d_fwd")
    pr.done()
    return y


def conv_fwd_mat(geom, x, w_fwd, bias, cin_pad, gb, gb_off, gb_st, st_off, act=ACT_LRELU, slope=0.2, aux=None, epi=EPI_STORE,
                 want_y=True):
    """conv (+ bias, + residual aux) followed by InstanceNorm + MAT modulation + activation of its output, one launch where
    the plane-resident kernel applies (s2p_conv2d_fwd_mat; conv_fwd + in_norm_fwd otherwise).
    Returns (y, y_mat, stats): the conv output (kept for the backward), the modulated activation, the norm statistics.
    want_y=False (a forward without a backward): where the launch is fused the conv output is not written at all (y is None)."""
    N, H, W, xp = x.shape
    Ho, Wo = geom.out_hw(H, W)
    C = geom.cout
    y_pitch = pad_to(C, chunk_elems(x.dtype))
    y_mat = torch.empty((N, Ho, Wo, C), dtype=x.dtype, device=x.device)
    stats = torch.empty(lib().s2p_in_stats_floats(N, Ho * Wo, C), dtype=torch.float32, device=x.device)
    d = geom.desc(x.dtype, N, H, W, cin_pad, xp, y_pitch)
    y = None
    if not SKIP_UNUSED_Y:
        want_y = True
    if not want_y:
        # (the answer is a property of the geometry: asked once -- the query walks the launcher's planning path on the host, which an
        # eager forward cannot afford six times per pass)
        # (a module-level table: layers compare their geometry objects attribute by attribute -- ConvLayer.wgrad_many -- so nothing
        # may be cached ON them)
        key = (geom.cin, geom.cout, geom.k, geom.stride, geom.pad, geom.transposed, geom.reflect, geom.groups, geom.output_padding,
               x.dtype, N, H, W, cin_pad, xp, gb is not None)
        if key not in _MAT_FUSED:
            _MAT_FUSED[key] = bool(lib().s2p_conv2d_mat_is_fused(ctypes.byref(d), 0, 1 if gb is not None else 0))
        want_y = not _MAT_FUSED[key]
    if want_y:
        y = torch.empty((N, Ho, Wo, y_pitch), dtype=x.dtype, device=x.device)
    need = lib().s2p_conv2d_fwd_workspace(ctypes.byref(d), epi)
    ws = torch.empty(need, dtype=torch.uint8, device=x.device) if need else None
    gbp, gb_pitch, stp, st_pitch = _gb_args(gb, gb_off, gb_st, st_off)
    pr = _Prof("fwd", geom, N, H, W, x.dtype)
    if pr.on:       # the fused launch also moves the norm's bytes: gamma, beta read + the modulated tensor written
        pr.rec["norm_bytes"] = float(N * Ho * Wo * C * x.element_size() * 3)
        pr.rec["variant"] = "mat_fwd"
    check(lib().s2p_conv2d_fwd_mat(ctypes.byref(d), ptr(x), ptr(w_fwd), ptr(bias), ptr(aux), ptr(y), epi, gbp, gb_pitch, stp,
                                   st_pitch, act, slope, IN_EPS, ptr(y_mat), C, ptr(stats), ptr(ws), need, stream()),
          "s2p_conv2d_fwd_mat")
    pr.done()
    return y, y_mat, stats


def conv_dgrad(geom, dy, w_bwd, x_shape, cin_pad, aux=None, epi=EPI_STORE, aux_act=ACT_NONE, slope=0.2, aux2=None):
    """dx of the conv (cudnn_convolution_backward_input equivalent).  For reflect-padded convs the padded-grid
    gradient is folded back (adjoint of F.pad(mode='reflect'))."""
    N, H, W, xp = x_shape
    d = geom.desc(dy.dtype, N, H, W, cin_pad, xp, dy.shape[3])
    if geom.reflect:
        p = geom.pad
        dxp = torch.empty((N, H + 2 * p, W + 2 * p, xp), dtype=dy.dtype, device=dy.device)
        need = lib().s2p_conv2d_dgrad_workspace(ctypes.byref(d))
        ws = torch.empty(need, dtype=torch.uint8, device=dy.device) if need else None
        pr = _Prof("dgrad", geom, N, H, W, dy.dtype)
        check(lib().s2p_conv2d_dgrad_ws(ctypes.byref(d), ptr(dy), ptr(w_bwd), None, None, ptr(dxp), EPI_STORE, ACT_NONE, 0.0,
                                        ptr(ws), need, stream()), "s2p_conv2d_dgrad")
        pr.done()
        dx = torch.empty((N, H, W, xp), dtype=dy.dtype, device=dy.device)
        check(lib().s2p_reflect_pad_bwd(dtype_id(dy.dtype), ptr(dxp), N, H, W, xp, p, ptr(dx), stream()),
              "s2p_reflect_pad_bwd")
        if aux2 is not None:
            dx = add(dx, aux2, out=dx)
        if epi == EPI_MUL_ACTGRAD:
            dx = act_bwd(dx, aux, aux_act, slope)
        elif epi == EPI_ADD:
            dx = add(dx, aux, out=dx)
        return dx
    dx = torch.empty((N, H, W, xp), dtype=dy.dtype, device=dy.device)
    if xp != cin_pad * geom.groups:
        dx.zero_()
    need = lib().s2p_conv2d_dgrad_workspace(ctypes.byref(d))
    ws = torch.empty(need, dtype=torch.uint8, device=dy.device) if need else None
    pr = _Prof("dgrad", geom, N, H, W, dy.dtype)
    check(lib().s2p_conv2d_dgrad_ws(ctypes.byref(d), ptr(dy), ptr(w_bwd), ptr(aux), ptr(aux2), ptr(dx), epi, aux_act, slope,
                                    ptr(ws), need, stream()), "s2p_conv2d_dgrad")
    pr.done()
    return dx


def conv_dgrad_mat(geom, dy, w_bwd, xn, cin_pad, stats, gb, gb_off, gb_st, st_off, act, slope, dgb, dgb_off, dgb_st, dst_off,
                   res=None, aux=None):
    """dgrad of a conv whose input was a MAT / InstanceNorm's output, fused with that norm's backward (s2p_conv2d_dgrad_mat):
    returns dL/d(xn) (+ res); writes d(gamma|beta) as in_bwd does.  xn: the norm's input [N,H,W,C].  aux: a second gradient
    arriving at the norm's output (layout of the dgrad result), added before the norm backward."""
    N, H, W, xp = xn.shape
    C = geom.cin
    if cin_pad != C or xp != C:
        # (a norm's channel count is a multiple of the 16-byte chunk -- the norm kernels require it -- so its tensor is dense; widths such
        # as --ndf 12 give 24 / 48 / 96 normed channels and take this path unchanged: tests/test_model_gpu.py::test_small_width_discriminator)
        raise ValueError("conv_dgrad_mat: the norm's tensor must be dense in channels (C %d, cin_pad %d, pitch %d)" % (C, cin_pad, xp))
    d = geom.desc(dy.dtype, N, H, W, cin_pad, C, dy.shape[3])
    dxn = torch.empty((N, H, W, C), dtype=dy.dtype, device=dy.device)
    fused = bool(lib().s2p_conv2d_mat_is_fused(ctypes.byref(d), 1, int(gb is not None))) and (aux is None or geom.k != 3)
    # scratch of the two-launch form only when that form runs
    d_mid = None if fused else torch.empty((N, H, W, C), dtype=dy.dtype, device=dy.device)
    sums = None if fused else torch.empty(lib().s2p_in_bwd_sums_floats(N, H * W, C), dtype=torch.float32, device=dy.device)
    need = lib().s2p_conv2d_dgrad_workspace(ctypes.byref(d))
    ws = torch.empty(need, dtype=torch.uint8, device=dy.device) if need else None
    gbp, gb_pitch, stp, st_pitch = _gb_args(gb, gb_off, gb_st, st_off)
    dgbp = dgb.data_ptr() + dgb_off * dgb.element_size() if dgb is not None else None
    dstp = dgb_st.data_ptr() + dst_off * 4 if dgb_st is not None else None
    _ = ptr(dgb), ptr(dgb_st)
    if res is not None:
        assert res.shape == dxn.shape and res.dtype == dxn.dtype and res.is_contiguous()
    if aux is not None:
        assert aux.shape == dxn.shape and aux.dtype == dxn.dtype and aux.is_contiguous()
    pr = _Prof("dgrad", geom, N, H, W, dy.dtype)
    if pr.on:       # + the norm backward's bytes: xn, gamma, beta read (+ res, aux), dxn, d(gamma), d(beta) written; dy is the conv's own
        nt = (2 + (res is not None) + (aux is not None)) if gb is None else (6 + (res is not None) + (aux is not None))
        pr.rec["norm_bytes"] = float(N * H * W * C * dy.element_size() * nt)
        pr.rec["variant"] = "mat_bwd"
    check(lib().s2p_conv2d_dgrad_mat(ctypes.byref(d), ptr(dy), ptr(w_bwd), ptr(d_mid), ptr(aux), ptr(xn), xp, ptr(stats), gbp, gb_pitch,
                                     stp, st_pitch, act, slope, IN_EPS, ptr(sums), ptr(dxn), C, dgbp,
                                     dgb.shape[3] if dgb is not None else 0, dstp,
                                     dgb_st.shape[1] if dgb_st is not None else 0, ptr(res), C if res is not None else 0,
                                     ptr(ws), need, stream()), "s2p_conv2d_dgrad_mat")
    pr.done()
    return dxn


def conv_wgrad(geom, x, dy, dw, cin_pad, cin_real, cout_real, dw_gstride=0, splitk=0, db=None):
    """dw += wgrad (cudnn_convolution_backward_weight equivalent); dw is an fp32 view in channels-last layout.
    The K-split partial sums go through a scratch buffer and are added in a fixed order (s2p_conv2d_wgrad_ws): no atomics."""
    N, H, W, xp = x.shape
    d = geom.desc(x.dtype, N, H, W, cin_pad, xp, dy.shape[3])
    need = lib().s2p_conv2d_wgrad_workspace(ctypes.byref(d), cin_real, cout_real)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=x.device)
    pr = _Prof("wgrad", geom, N, H, W, x.dtype)
    check(lib().s2p_conv2d_wgrad_ws(ctypes.byref(d), ptr(x), ptr(dy), ptr(dw), ptr(db), cin_real, cout_real, dw_gstride, splitk,
                                    ptr(ws), need, stream()), "s2p_conv2d_wgrad_ws")
    pr.done()


def conv_wgrad_batched(geom, jobs, cin_pad, cin_real, cout_real):
    """dw_j += wgrad(x_j, dy_j), db_j += sum dy_j for several convs of the SAME geometry in one launch
    (s2p_conv2d_wgrad_batched).  jobs: list of (x, x_ch_off, dy, dy_ch_off, dw, db): x / dy NHWC tensors (all the
    same shape) read from channel offset *_ch_off (a grouped conv is one job per group), dw fp32 view, db fp32 view or
    None.  `geom` must describe ONE group (groups == 1)."""
    x0, _, dy0 = jobs[0][0], jobs[0][1], jobs[0][2]
    N, H, W, xp = x0.shape
    d = geom.desc(x0.dtype, N, H, W, cin_pad, xp, dy0.shape[3])
    arr = (_lib.WgradJob * len(jobs))()
    esz = x0.element_size()
    for i, (x, xo, dy, dyo, dw, db) in enumerate(jobs):
        assert x.shape == x0.shape and dy.shape == dy0.shape
        arr[i] = _lib.WgradJob(ptr(x) + xo * esz, ptr(dy) + dyo * esz, ptr(dw), ptr(db))
    need = lib().s2p_conv2d_wgrad_batched_workspace(ctypes.byref(d), len(jobs), cin_real, cout_real)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=x0.device)
    pr = _Prof("wgrad", geom, N, H, W, x0.dtype)
    if pr.on:
        pr.rec["flops"] *= len(jobs)
        pr.rec["shape"] = pr.rec["shape"][:7] + (len(jobs),) + pr.rec["shape"][8:]
    check(lib().s2p_conv2d_wgrad_batched(ctypes.byref(d), arr, len(jobs), cin_real, cout_real, ptr(ws), need, stream()),
          "s2p_conv2d_wgrad_batched")
    pr.done()


def channel_sum(dy, C, db):
    """db += dy.sum over pixels (bias gradient)."""
    pixels = dy.shape[0] * dy.shape[1] * dy.shape[2]
    check(lib().s2p_channel_sum(dtype_id(dy.dtype), ptr(dy), pixels, C, dy.shape[3], ptr(db), stream()),
          "s2p_channel_sum")


def in_stats(x, C):
    """Per-(n,c) InstanceNorm statistics: an opaque buffer of partial moments (include/s2p_hip.h) consumed by
    in_apply_fwd / in_bwd with the same x shape."""
    N, H, W, xp = x.shape
    stats = torch.empty(lib().s2p_in_stats_floats(N, H * W, C), dtype=torch.float32, device=x.device)
    pr = _ProfNorm("norm_stats", x, C, N * H * W * C * x.element_size(), False)
    check(lib().s2p_in_stats(dtype_id(x.dtype), ptr(x), N, H * W, C, xp, IN_EPS, ptr(stats), stream()), "s2p_in_stats")
    pr.done()
    return stats


def _gb_args(gb, gb_off, gb_st, st_off):
    gbp = gb.data_ptr() + gb_off * gb.element_size() if gb is not None else None
    gb_pitch = gb.shape[3] if gb is not None else 0
    stp = gb_st.data_ptr() + st_off * 4 if gb_st is not None else None
    st_pitch = gb_st.shape[1] if gb_st is not None else 0
    return gbp, gb_pitch, stp, st_pitch


def in_apply_fwd(x, C, stats, gb=None, gb_off=0, gb_st=None, st_off=0, act=ACT_NONE, slope=0.2):
    """F.instance_norm(x) * (1 + gamma) + beta, then activation; gamma/beta = image map + per-sample state affine."""
    N, H, W, xp = x.shape
    y = torch.empty((N, H, W, C), dtype=x.dtype, device=x.device)
    gbp, gb_pitch, stp, st_pitch = _gb_args(gb, gb_off, gb_st, st_off)
    pr = _ProfNorm("norm_fwd", x, C, N * H * W * C * x.element_size() * (4 if gb is not None else 2), gb is not None)
    check(lib().s2p_in_apply_fwd(dtype_id(x.dtype), ptr(x), N, H * W, C, xp, ptr(stats), gbp, gb_pitch, stp, st_pitch,
                                 act, slope, IN_EPS, ptr(y), C, stream()), "s2p_in_apply_fwd")
    pr.done()
    return y


def in_norm_fwd(x, C, gb=None, gb_off=0, gb_st=None, st_off=0, act=ACT_NONE, slope=0.2):
    """in_stats + in_apply_fwd as ONE call (one fused launch for small planes): returns (y, stats)."""
    N, H, W, xp = x.shape
    stats = torch.empty(lib().s2p_in_stats_floats(N, H * W, C), dtype=torch.float32, device=x.device)
    y = torch.empty((N, H, W, C), dtype=x.dtype, device=x.device)
    gbp, gb_pitch, stp, st_pitch = _gb_args(gb, gb_off, gb_st, st_off)
    esz = x.element_size()
    pr = _ProfNorm("norm_fwd", x, C, N * H * W * C * esz * ((4 if gb is not None else 2) + 1), gb is not None)
    check(lib().s2p_in_norm_fwd(dtype_id(x.dtype), ptr(x), N, H * W, C, xp, gbp, gb_pitch, stp, st_pitch, act, slope, IN_EPS,
                                ptr(y), C, ptr(stats), stream()), "s2p_in_norm_fwd")
    pr.done()
    return y, stats


def in_bwd(da, x, C, stats, gb=None, gb_off=0, gb_st=None, st_off=0, act=ACT_NONE, slope=0.2, dgb=None, dgb_off=0,
           dgb_st=None, dst_off=0, res=None):
    """Backward of in_apply_fwd.  Returns dx; writes d(gamma_img|beta_img) into dgb at channel offset dgb_off and
    d(gamma_st|beta_st) into the fp32 [N, pitch] tensor dgb_st at column offset dst_off (layout of gb_st).
    res: optional [N,H,W,C] tensor added to dx in the same launch (the skip-connection gradient of a residual block)."""
    N, H, W, xp = x.shape
    sums = torch.empty(lib().s2p_in_bwd_sums_floats(N, H * W, C), dtype=torch.float32, device=x.device)
    gbp, gb_pitch, stp, st_pitch = _gb_args(gb, gb_off, gb_st, st_off)
    dt = dtype_id(x.dtype)
    el = N * H * W * C * x.element_size()
    # reduce reads x, da (+gamma, beta); apply reads the same and writes dx (+dgamma, dbeta)
    pr = _ProfNorm("norm_bwd", x, C, el * ((4 + 4 + 3) if gb is not None else (2 + 2 + 1)), gb is not None)
    dx = torch.empty((N, H, W, C), dtype=x.dtype, device=x.device)
    dgbp = dgb.data_ptr() + dgb_off * dgb.element_size() if dgb is not None else None
    dstp = dgb_st.data_ptr() + dst_off * 4 if dgb_st is not None else None
    _ = ptr(dgb), ptr(dgb_st)
    # one fused launch for small planes, the reduce + apply pair otherwise (s2p_in_norm_bwd decides)
    if res is not None:
        assert res.shape == dx.shape and res.dtype == dx.dtype and res.is_contiguous()
    check(lib().s2p_in_norm_bwd_res(dt, ptr(da), da.shape[3], ptr(x), N, H * W, C, xp, ptr(stats), gbp, gb_pitch, stp,
                                    st_pitch, act, slope, IN_EPS, ptr(sums), ptr(dx), C, dgbp,
                                    dgb.shape[3] if dgb is not None else 0, dstp,
                                    dgb_st.shape[1] if dgb_st is not None else 0, ptr(res), C if res is not None else 0,
                                    stream()), "s2p_in_norm_bwd")
    pr.done()
    return dx


def linear_fwd(x, w_fwd, bias, K, N, act=ACT_NONE, slope=0.2, n_store=None):
    """y = act(x @ w.T + b) for the small fp32 layers of the state path.  x: fp32 [M, x_pitch]; w_fwd: packed
    [1][N][1][Kpad] fp32 (ParamStore pack); returns fp32 [M, n_store] (columns >= N zero)."""
    M, xp = x.shape
    n_store = pad_to(N, 4) if n_store is None else n_store
    y = torch.empty((M, n_store), dtype=torch.float32, device=x.device)
    check(lib().s2p_linear_fwd(ptr(x), M, K, xp, ptr(w_fwd), w_fwd.shape[-1], ptr(bias), N, act, slope, ptr(y), n_store,
                               n_store, stream()), "s2p_linear_fwd")
    return y


def linear_bwd(x, dy, y, w_bwd, K, k_real, N, act, slope, dw, db, need_dx=True):
    """Backward of linear_fwd: dw += dpre.T @ x, db += dpre.sum(0), returns dx = dpre @ w (or None); dpre = dy * act'(y)."""
    M, xp = x.shape
    dx = torch.empty((M, xp), dtype=torch.float32, device=x.device) if need_dx else None
    need = lib().s2p_linear_bwd_workspace(M, K, N) if need_dx else 0
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=x.device)
    check(lib().s2p_linear_bwd(ptr(x), xp, ptr(dy), dy.shape[1], ptr(y), y.shape[1] if y is not None else 0, M, K, k_real, N,
                               ptr(w_bwd), w_bwd.shape[-1] if w_bwd is not None else 0, act, slope, ptr(dw), k_real, ptr(db),
                               ptr(dx), xp, ptr(ws), need, stream()), "s2p_linear_bwd")
    return dx


def posenc(state, L, pitch):
    N, S = state.shape
    out = torch.empty((N, pitch), dtype=torch.float32, device=state.device)
    check(lib().s2p_posenc_fwd(ptr(state), N, S, L, ptr(out), pitch, stream()), "s2p_posenc_fwd")
    return out


def avgpool_fwd(x):
    N, H, W, C = x.shape
    y = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), dtype=x.dtype, device=x.device)
    check(lib().s2p_avgpool3x3s2_fwd(dtype_id(x.dtype), ptr(x), N, H, W, C, ptr(y), stream()), "s2p_avgpool3x3s2_fwd")
    return y


def avgpool_bwd(dy, x_shape, dx=None, accumulate=False):
    N, H, W, C = x_shape
    if dx is None:
        dx = torch.empty(x_shape, dtype=dy.dtype, device=dy.device)
    check(lib().s2p_avgpool3x3s2_bwd(dtype_id(dy.dtype), ptr(dy), N, H, W, C, ptr(dx), int(accumulate), stream()),
          "s2p_avgpool3x3s2_bwd")
    return dx


def maxpool_fwd(x):
    N, H, W, C = x.shape
    y = torch.empty((N, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
    check(lib().s2p_maxpool2x2_fwd(dtype_id(x.dtype), ptr(x), N, H, W, C, ptr(y), stream()), "s2p_maxpool2x2_fwd")
    return y


def maxpool_bwd(dy, x):
    N, H, W, C = x.shape
    dx = torch.empty_like(x)
    check(lib().s2p_maxpool2x2_bwd(dtype_id(x.dtype), ptr(dy), ptr(x), N, H, W, C, ptr(dx), stream()),
          "s2p_maxpool2x2_bwd")
    return dx


def resize_nearest(x, Ho, Wo):
    N, H, W, C = x.shape
    y = torch.empty((N, Ho, Wo, C), dtype=x.dtype, device=x.device)
    check(lib().s2p_resize_nearest(dtype_id(x.dtype), ptr(x), N, H, W, C, ptr(y), Ho, Wo, stream()), "s2p_resize_nearest")
    return y


def nchw_to_nhwc(x, dtype, pitch, out=None, c_off=0, zero_pad=True):
    """fp32 NCHW image -> NHWC compute-dtype tensor with zero-padded channel pitch."""
    N, C, H, W = x.shape
    x = x.contiguous()
    y = out if out is not None else torch.empty((N, H, W, pitch), dtype=dtype, device=x.device)
    check(lib().s2p_nchw_to_nhwc(dtype_id(dtype), ptr(x), N, C, H, W, ptr(y), pitch, c_off, int(zero_pad), stream()),
          "s2p_nchw_to_nhwc")
    return y


def nhwc_to_nchw(x, C, c_off=0, out=None, accumulate=False):
    N, H, W, xp = x.shape
    y = out if out is not None else torch.empty((N, C, H, W), dtype=torch.float32, device=x.device)
    if not y.is_contiguous() or tuple(y.shape) != (N, C, H, W) or y.dtype != torch.float32:
        raise ValueError("nhwc_to_nchw: `out` must be a contiguous fp32 [N,C,H,W] tensor, got shape %s strides %s %s" % (
            tuple(y.shape), y.stride(), y.dtype))
    check(lib().s2p_nhwc_to_nchw(dtype_id(x.dtype), ptr(x), xp, c_off, N, C, H, W, ptr(y), int(accumulate), stream()),
          "s2p_nhwc_to_nchw")
    return y


def cast(x, dtype):
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    check(lib().s2p_cast(dtype_id(x.dtype), ptr(x), dtype_id(dtype), ptr(y), x.numel(), stream()), "s2p_cast")
    return y


def act_bwd(dy, y, act, slope=0.2):
    dx = torch.empty_like(dy)
    check(lib().s2p_act_bwd(dtype_id(dy.dtype), ptr(dy), ptr(y), dy.numel(), act, slope, ptr(dx), stream()), "s2p_act_bwd")
    return dx


def scale_(x, scale_dev):
    """x *= scale (fp32 device scalar)."""
    check(lib().s2p_scale(dtype_id(x.dtype), ptr(x), x.numel(), ptr(scale_dev), stream()), "s2p_scale")
    return x


def l1_loss(a, b, scale, loss_out, grad_a=None, accumulate=False):
    """loss_out += scale * sum|a-b|; grad_a (+)= scale*sign(a-b).  a, b contiguous, same numel."""
    check(lib().s2p_l1_loss(dtype_id(a.dtype), ptr(a), ptr(b), a.numel(), scale, ptr(loss_out), ptr(grad_a),
                            int(accumulate), stream()), "s2p_l1_loss")


def l1_loss_multi(jobs):
    """Several L1 terms in one launch.  jobs: list of (a, b, scale, loss_out, grad_a|None) with a, b contiguous NHWC
    activations of one dtype; loss_out += scale * sum|a-b|, grad_a = scale * sign(a-b).  Lists longer than the kernel's job
    table go out in several launches."""
    cap = 16
    for i0 in range(0, len(jobs), cap):
        part = jobs[i0:i0 + cap]
        arr = (_lib.L1Job * len(part))()
        for i, (a, b, scale, loss_out, g) in enumerate(part):
            assert a.dtype == part[0][0].dtype and a.numel() == b.numel()
            arr[i] = _lib.L1Job(ptr(a), ptr(b), ptr(g), a.numel(), scale, ptr(loss_out))
        check(lib().s2p_l1_loss_multi(dtype_id(part[0][0].dtype), arr, len(part), stream()), "s2p_l1_loss_multi")


def hinge_loss(x, count, mode, scale, loss_out, grad_x=None, x_off=0):
    esz = x.element_size()
    xp = x.data_ptr() + x_off * esz
    gp = grad_x.data_ptr() + x_off * esz if grad_x is not None else None
    _ = ptr(x)
    check(lib().s2p_hinge_loss(dtype_id(x.dtype), xp, count, mode, scale, ptr(loss_out), gp, stream()), "s2p_hinge_loss")


def hinge_loss_nhwc(x, mode, scale, loss_out, want_grad=True):
    """Hinge term on an NHWC logit map [B,h,w,pitch] (channel 0 = logit): loss_out += ..., returns the gradient in the
    same layout (padding channels zero) or None."""
    B, h, w, pitch = x.shape
    g = torch.empty_like(x) if want_grad else None
    check(lib().s2p_hinge_loss_strided(dtype_id(x.dtype), ptr(x), B * h * w, pitch, mode, scale, ptr(loss_out), ptr(g), stream()),
          "s2p_hinge_loss_strided")
    return g


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    check(lib().s2p_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, step, grad_scale,
                              stream()), "s2p_adam_step")


def add(a, b, out=None):
    out = out if out is not None else torch.empty_like(a)
    check(lib().s2p_add(dtype_id(a.dtype), ptr(a), ptr(b), ptr(out), a.numel(), stream()), "s2p_add")
    return out


def copy_channels(src, src_off, dst, dst_off, C, accumulate=False, src_rows=None, dst_row0=0, src_row0=0):
    """dst[dst_row0 + p, ..., dst_off:dst_off+C] (+)= src[src_row0 + p, ..., src_off:src_off+C] over `src_rows` images."""
    n_img = src.shape[0] if src_rows is None else src_rows
    ppi = src.shape[1] * src.shape[2]
    sp, dp = src.shape[3], dst.shape[3]
    sptr = src.data_ptr() + src_row0 * ppi * sp * src.element_size()
    dptr = dst.data_ptr() + dst_row0 * ppi * dp * dst.element_size()
    _ = ptr(src), ptr(dst)
    check(lib().s2p_copy_channels(dtype_id(src.dtype), sptr, sp, src_off, dptr, dp, dst_off, C, n_img * ppi,
                                  int(accumulate), stream()), "s2p_copy_channels")
    return dst


def adam_step_dev(p, g, m, v, lr, beta1, beta2, eps, step_dev, grad_scale=1.0):
    """Graph-capturable Adam: `step_dev` is an int32 device tensor incremented on the device."""
    check(lib().s2p_adam_step_dev(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, ptr(step_dev),
                                  grad_scale, stream()), "s2p_adam_step_dev")


def adam_step_dev_part(p, g, m, v, lr, beta1, beta2, eps, step_dev, grad_scale=1.0, tick=True):
    """One optimizer step applied to a RANGE (views of the flat buffers); the first part ticks the device step counter."""
    check(lib().s2p_adam_step_dev_part(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, ptr(step_dev),
                                       grad_scale, int(tick), stream()), "s2p_adam_step_dev_part")


def u8_to_nhwc(x_u8, dtype, pitch):
    """uint8 NHWC frames [N,H,W,C] on the device -> compute-dtype NHWC in [-1,1], zero-padded to `pitch` channels."""
    N, H, W, C = x_u8.shape
    y = torch.empty((N, H, W, pitch), dtype=dtype, device=x_u8.device)
    check(lib().s2p_u8_to_nhwc(dtype_id(dtype), ptr(x_u8), N * H * W, C, ptr(y), pitch, stream()), "s2p_u8_to_nhwc")
    return y


def nhwc_to_u8(x, C, out=None):
    """compute-dtype NHWC in [-1,1] -> uint8 NHWC [N,H,W,C] (the on-disk layout of image_observations*)."""
    N, H, W, xp = x.shape
    y = out if out is not None else torch.empty((N, H, W, C), dtype=torch.uint8, device=x.device)
    check(lib().s2p_nhwc_to_u8(dtype_id(x.dtype), ptr(x), xp, N * H * W, C, ptr(y), stream()), "s2p_nhwc_to_u8")
    return y
