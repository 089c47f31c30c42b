"""Diagnose the run-to-run differences of the state-path gradients: record the operands of every linear_bwd call of two
identical G steps (clones taken on the stream the call runs on) and report which ones differ."""
import os, sys, io, contextlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import ops
from s2p_amd.options.train_options import TrainOptions
from s2p_amd.trainers.pix2pix_trainer import Pix2PixTrainer
B = int(os.environ.get("BATCH", "64"))
opt = TrainOptions().parse(["--env_type", "cheetah", "--batchSize", str(B), "--precision", "bf16", "--gpu_ids", "0",
                            "--checkpoints_dir", "/tmp/repro_ck"], quiet=True)
with contextlib.redirect_stdout(io.StringIO()):
    tr = Pix2PixTrainer(opt)
m = tr.pix2pix_model
g = torch.Generator().manual_seed(0)
data = dict(prev_image=(torch.rand(B, 3, 84, 84, generator=g) * 2 - 1).cuda(), image=(torch.rand(B, 3, 84, 84, generator=g) * 2 - 1).cuda(),
            state=torch.randn(B, 17, generator=g).cuda())
rec = []
orig = ops.linear_bwd
import ctypes
from s2p_amd import _lib
def with_ws(x, dy, y, w_bwd, K, k_real, N, act, slope, dw, db, e):
    """ops.linear_bwd with the split-K workspace kept: e["part"] = clone of the partial tiles after the call"""
    M, xp = x.shape
    dx = torch.empty((M, xp), dtype=torch.float32, device=x.device)
    need = _lib.lib().s2p_linear_bwd_workspace(M, K, N)
    ws = torch.full((max(need, 16),), 0x7f, dtype=torch.uint8, device=x.device)       # poisoned
    _lib.check(_lib.lib().s2p_linear_bwd(_lib.ptr(x), xp, _lib.ptr(dy), dy.shape[1], _lib.ptr(y), y.shape[1] if y is not None else 0, M, K, k_real, N,
                               _lib.ptr(w_bwd), w_bwd.shape[-1] if w_bwd is not None else 0, act, slope, _lib.ptr(dw), k_real, _lib.ptr(db),
                               _lib.ptr(dx), xp, _lib.ptr(ws), need, _lib.stream()), "s2p_linear_bwd")
    e["part"] = ws.clone().view(torch.float32)
    return dx
def spy(x, dy, y, w_bwd, K, k_real, N, act, slope, dw, db, need_dx=True):
    e = dict(x=x.clone(), dy=dy.clone(), y=None if y is None else y.clone(), dw0=dw.clone())
    if N >= 2048 and need_dx:
        dx = with_ws(x, dy, y, w_bwd, K, k_real, N, act, slope, dw, db, e)
    else:
        dx = orig(x, dy, y, w_bwd, K, k_real, N, act, slope, dw, db, need_dx)
    e["dw1"] = dw.clone(); e["dx"] = None if dx is None else dx.clone()
    rec.append(e)
    return dx
ops.linear_bwd = spy
import s2p_amd.models.networks.generator as G
runs = []
for r in range(2):
    rec.clear()
    tr.optimizer_G.zero_grad()
    L, _ = m(data, mode="generator"); tr._backward(L)
    torch.cuda.synchronize()
    runs.append([dict(e) for e in rec])
for i, (a, b) in enumerate(zip(*runs)):
    for k in a:
        if a[k] is None: continue
        if not torch.equal(a[k], b[k]):
            d = (a[k].double() - b[k].double())
            nz = d.abs() > 0
            rows = nz.any(dim=1).sum().item() if d.dim() == 2 else -1
            cols = nz.any(dim=0).sum().item() if d.dim() == 2 else -1
            print("call %d %-4s shape %s differs: rel-L2 %.2e, %d elements in %d rows / %d cols" % (
                i, k, tuple(a[k].shape), float(d.norm() / (a[k].double().norm() + 1e-30)), int(nz.sum()), rows, cols))
a, b = runs[0][0], runs[1][0]
W = m.netG.lay["fc_state"].pk.w_bwd.view(256, -1).double()
ref = a["dy"].double() @ W[:, :a["dy"].shape[1]].t()
for nm, r in (("run0", a), ("run1", b)):
    d = (r["dx"].double() - ref).abs()
    badmask = d > 1e-3 * ref.abs().max()
    rows = badmask.any(1).nonzero().flatten().tolist(); cols = badmask.any(0).nonzero().flatten().tolist()
    print(nm, "dx vs float64: max err %.3e; bad rows %s cols %s" % (float(d.max()), rows[:3] + ["..."] + rows[-3:] if rows else [], cols[:3] + ["..."] + cols[-3:] if cols else []))
    if rows:
        i, j = rows[0], cols[0]
        print("   sample [%d,%d]: got %.6f ref %.6f ; got-ref %.6f" % (i, j, float(r["dx"][i, j]), float(ref[i, j]), float(r["dx"][i, j]) - float(ref[i, j])))
        # is the error one missing / doubled K split (512 wide)?
        dyr, Wr = r["dy"][i].double(), W[j, :r["dy"].shape[1]]
        parts = [(dyr[z * 512:(z + 1) * 512] * Wr[z * 512:(z + 1) * 512]).sum().item() for z in range(12)]
        print("   per-split partials:", ["%.4f" % p for p in parts])
pk = m.netG.lay["fc_state"].pk
torch.cuda.synchronize()
dws = torch.zeros_like(a["dw0"])
dxc = orig(a["x"], a["dy"], None, pk.w_bwd, pk.Cpad, pk.C, pk.R, 0, 0.0, dws, torch.zeros(pk.R, device="cuda"))
torch.cuda.synchronize()
print("standalone re-run on the recorded operands: equals run0 %s, equals run1 %s" % (torch.equal(dxc, a["dx"]), torch.equal(dxc, b["dx"])))
nz = (a["dx"] != b["dx"]).nonzero()
for i, j in nz[:6].tolist():
    print("   [%d,%d] run0 %.9e run1 %.9e standalone %.9e ref64 %.9e" % (i, j, float(a["dx"][i, j]), float(b["dx"][i, j]), float(dxc[i, j]), float(ref[i, j])))
pa, pb = a["part"].view(12, 64, 256), b["part"].view(12, 64, 256)
pref = torch.stack([a["dy"][:, z * 512:(z + 1) * 512].double() @ W[:, z * 512:(z + 1) * 512].t() for z in range(12)])
for z in range(12):
    da, db_ = (pa[z].double() - pref[z]).abs(), (pb[z].double() - pref[z]).abs()
    na, nb = int((da > 1e-12).sum()), int((db_ > 1e-12).sum())
    if na or nb:
        bad = (da > 1e-12) if na else (db_ > 1e-12)
        print("   split %2d: run0 %d / run1 %d elements off; rows %s cols %s" % (z, na, nb, bad.any(1).nonzero().flatten().tolist()[:6], bad.any(0).nonzero().flatten().tolist()[:6]))
print("dy: absmax %.3e, rms %.3e; dx rms %.3e" % (float(a["dy"].abs().max()), float(a["dy"].pow(2).mean().sqrt()), float(a["dx"].pow(2).mean().sqrt())))
print("done: %d linear_bwd calls per run" % len(runs[0]))
