"""Diagnostic (GPU box): where does the fp32 HIP generator lose precision against the float64 oracle?
Per intermediate: relative L2 / max error of HIP fp32 and of the torch-fp32 oracle, and the number of activation-mask
disagreements (elements whose pre-activation sign differs from float64).  Then the gradient error with and without
mask control (oracle run with the masks the HIP forward took)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import s2p_oracle as O
from s2p_amd import ops
from s2p_amd._lib import chunk_elems
from test_model_gpu import build, make_inputs, to64, grad_errors


def nchw(y, C):
    return y[..., :C].float().cpu().permute(0, 3, 1, 2).contiguous()


def rl2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def rmax(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-300))


def hip_trace(net, prev, state):
    dt = net.compute_dtype
    img = ops.nchw_to_nhwc(prev.float(), dt, chunk_elems(dt))
    out, c = net.fwd_nhwc(img, state.float(), save=True)
    t = {}
    names = ["stem"] + [f"down{i}" for i in range(net.n_down)]
    for nm, (xin, x, s, a) in zip(names, c["enc"]):
        t[nm + ".conv"] = nchw(x, x.shape[3]); t[nm + ".post"] = nchw(a, a.shape[3])
    C = net.c_mid
    for b, (x, sA, nA, c0, sB, nB) in enumerate(c["blocks"]):
        t[f"blocks.{b}.norm_0.post"] = nchw(nA, C); t[f"blocks.{b}.conv_0"] = nchw(c0, C)
        t[f"blocks.{b}.norm_1.post"] = nchw(nB, C)
        if b > 0:
            t[f"blocks.{b - 1}.out"] = nchw(x, C)
    for i, (xin, u, s, x) in enumerate(c["dec"]):
        if i == 0:
            t[f"blocks.{net.n_blocks - 1}.out"] = nchw(xin, C)
        t[f"up{i}.conv"] = nchw(u, u.shape[3]); t[f"up{i}.post"] = nchw(x, x.shape[3])
    t["w"] = c["hs"][-1].view(prev.shape[0], -1).float().cpu()
    t["out"] = nchw(out, 3)
    return t


def generator_masks_of(model, y):
    from test_model_gpu import generator_masks
    return generator_masks(model.netG, y.grad_fn.next_functions[0][0].c)


def main():
    import tempfile
    opt, model, spec, pg, pd, pv = build("fp32", tempfile.mkdtemp())
    prev, state, real = make_inputs(2, 84, 84, 17)
    th = hip_trace(model.netG, prev.cuda(), state.cuda())
    torch.cuda.synchronize()
    t64, t32 = {}, {}
    pg64 = {k: v.double() for k, v in pg.items()}
    y64 = O.generator_forward(pg64, prev.double(), state.double(), spec, trace=t64)
    y32 = O.generator_forward(pg, prev, state, spec, trace=t32)
    t64["out"], t32["out"] = y64, y32
    print(f"{'tensor':28s} {'hip relL2':>10s} {'hip max':>10s} {'t32 relL2':>10s} {'flips hip':>9s} {'flips t32':>9s} {'numel':>9s}")
    for k in t64:
        ref = t64[k]
        post = k + ".post"
        if k in th:
            print(f"{k:28s} {rl2(th[k], ref):10.2e} {rmax(th[k], ref):10.2e} {rl2(t32[k], ref):10.2e}")
        if post in th:
            act = torch.relu(ref) if ("norm" not in k) else torch.nn.functional.leaky_relu(ref, 0.2)
            fh = int(((th[post] > 0) != (ref > 0)).sum()); f32 = int(((t32[k] > 0) != (ref > 0)).sum())
            print(f"{post:28s} {rl2(th[post], act):10.2e} {rmax(th[post], act):10.2e} {'':10s} {fh:9d} {f32:9d} {ref.numel():9d}")
    # gradients: plain vs mask-controlled
    y = model.netG(prev.cuda(), state.cuda())
    masks = generator_masks_of(model, y)
    r = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
    model.netG.store.zero_grad()
    (y * r.cuda()).sum().backward()
    torch.cuda.synchronize()
    p64 = to64(pg)
    (O.generator_forward(p64, prev.double(), state.double(), spec) * r.double()).sum().backward()
    e_plain = grad_errors(dict(model.netG.named_parameters()), p64)
    p64m = to64(pg)
    (O.generator_forward(p64m, prev.double(), state.double(), spec, masks=masks) * r.double()).sum().backward()
    e_mask = grad_errors(dict(model.netG.named_parameters()), p64m)
    ws = sorted(e_mask.items(), key=lambda kv: -kv[1])
    print("mask-controlled worst:", [(k, f"{e:.2e}") for k, e in ws[:12]])
    print("mask-controlled median:", f"{sorted(e_mask.values())[len(e_mask) // 2]:.2e}",
          " plain worst:", f"{max(e_plain.values()):.2e}", " plain median:", f"{sorted(e_plain.values())[len(e_plain) // 2]:.2e}")


if __name__ == "__main__":
    main()
