"""CPU oracle for the S2P hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

PARITY UNPINNED: `/root/reference` (dsshim0125/s2p @ v1) does not contain the
generator / discriminator / loss code, has no tests and no
golden vectors.  This file is therefore a plain
`torch.nn.functional` fp32 restatement of the frozen spec in `SPEC.md`, which is
in turn constrained only by the prose the reference does ship:

  * inputs (previous image, state) -> next image      rebuttal.md:6, :71, :154
  * MAT block: AdaIN-style modulation from state AND image  rebuttal.md:146-154
  * losses L1 + GAN + perceptual (ImageNet VGG)       rebuttal.md:71, :135, :187-188
  * CLI / checkpoint surface                           README.md:19-33, :59
  * lineage NVlabs/SPADE, nerf-pytorch posenc, StyleGAN mapping  README.md:72-75

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module.  The product package `s2p_amd` never does.

Everything is a pure function of a flat ``{name: tensor}`` parameter dict with the
same key names and logical shapes as the product modules' ``state_dict()``.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

LRELU = 0.2
IN_EPS = 1e-5


class Spec:
    """Frozen hyper-parameters (SPEC.md).  Mirrors the product option names."""

    def __init__(self, state_dim=17, ngf=64, n_down=2, n_blocks=6, nhidden=128,
                 w_dim=256, posenc_L=10, n_mlp=4, ndf=64, num_D=2, n_layers_D=4,
                 lambda_feat=10.0, lambda_vgg=10.0, lambda_l1=10.0):
        self.state_dim = state_dim
        self.ngf = ngf
        self.n_down = n_down
        self.n_blocks = n_blocks
        self.nhidden = nhidden
        self.w_dim = w_dim
        self.posenc_L = posenc_L
        self.n_mlp = n_mlp
        self.ndf = ndf
        self.num_D = num_D
        self.n_layers_D = n_layers_D
        self.lambda_feat = lambda_feat
        self.lambda_vgg = lambda_vgg
        self.lambda_l1 = lambda_l1

    @property
    def posenc_dim(self):
        return self.state_dim * (1 + 2 * self.posenc_L)


# --------------------------------------------------------------------------- #
# parameter construction (deterministic, seeded)                               #
# --------------------------------------------------------------------------- #
def _xavier_normal(shape, gain, gen, transposed=False):
    # torch.nn.init.xavier_normal_ semantics: fan_in = size(1)*rf, fan_out = size(0)*rf
    rf = 1
    for s in shape[2:]:
        rf *= s
    fan_in, fan_out = shape[1] * rf, shape[0] * rf
    std = gain * math.sqrt(2.0 / (fan_in + fan_out))
    return torch.randn(shape, generator=gen, dtype=torch.float32) * std


def generator_param_shapes(spec):
    s = OrderedDict()
    d = spec.posenc_dim
    for i in range(spec.n_mlp):
        s[f"state_map.fc{i}.weight"] = (spec.w_dim, d if i == 0 else spec.w_dim)
        s[f"state_map.fc{i}.bias"] = (spec.w_dim,)
    s["stem.weight"] = (spec.ngf, 3, 7, 7)
    c = spec.ngf
    for i in range(spec.n_down):
        s[f"down{i}.weight"] = (2 * c, c, 3, 3)
        c *= 2
    for b in range(spec.n_blocks):
        for j in range(2):
            p = f"blocks.{b}.norm_{j}"
            s[p + ".mlp_shared.weight"] = (spec.nhidden, 3, 3, 3)
            s[p + ".mlp_shared.bias"] = (spec.nhidden,)
            s[p + ".mlp_gamma.weight"] = (c, spec.nhidden, 3, 3)
            s[p + ".mlp_gamma.bias"] = (c,)
            s[p + ".mlp_beta.weight"] = (c, spec.nhidden, 3, 3)
            s[p + ".mlp_beta.bias"] = (c,)
            s[p + ".fc_state.weight"] = (2 * c, spec.w_dim)
            s[p + ".fc_state.bias"] = (2 * c,)
            s[f"blocks.{b}.conv_{j}.weight"] = (c, c, 3, 3)
            s[f"blocks.{b}.conv_{j}.bias"] = (c,)
    for i in range(spec.n_down):
        s[f"up{i}.weight"] = (c, c // 2, 3, 3)      # ConvTranspose2d layout [Cin, Cout, kh, kw]
        c //= 2
    s["out.weight"] = (3, c, 7, 7)
    s["out.bias"] = (3,)
    return s


def discriminator_param_shapes(spec):
    s = OrderedDict()
    for k in range(spec.num_D):
        nf = spec.ndf
        p = f"discriminator_{k}"
        s[f"{p}.model0.weight"] = (nf, 6, 4, 4)
        s[f"{p}.model0.bias"] = (nf,)
        for n in range(1, spec.n_layers_D):
            nf_prev, nf = nf, min(nf * 2, 512)
            s[f"{p}.model{n}.weight"] = (nf, nf_prev, 4, 4)   # norm follows: no bias
        s[f"{p}.model{spec.n_layers_D}.weight"] = (1, nf, 4, 4)
        s[f"{p}.model{spec.n_layers_D}.bias"] = (1,)
    return s


VGG_CFG = [  # (name, cin, cout); 'P' = 2x2 max-pool.  torchvision vgg19.features[0:30)
    ("conv1_1", 3, 64), ("conv1_2", 64, 64), "P",
    ("conv2_1", 64, 128), ("conv2_2", 128, 128), "P",
    ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256), ("conv3_4", 256, 256), "P",
    ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512), ("conv4_4", 512, 512), "P",
    ("conv5_1", 512, 512),
]
VGG_TAPS = ("conv1_1", "conv2_1", "conv3_1", "conv4_1", "conv5_1")
VGG_WEIGHTS = (1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0)


def vgg_param_shapes():
    s = OrderedDict()
    for item in VGG_CFG:
        if item == "P":
            continue
        name, cin, cout = item
        s[f"{name}.weight"] = (cout, cin, 3, 3)
        s[f"{name}.bias"] = (cout,)
    return s


def init_params(shapes, seed, gain=0.02, kaiming=False):
    """xavier-normal(gain) weights, zero biases (SPADE `init_weights('xavier', 0.02)`
    convention [UPSTREAM-RECALL]); `kaiming=True` gives He-normal, used for the
    seeded stand-in VGG so its activations do not collapse."""
    gen = torch.Generator().manual_seed(seed)
    p = OrderedDict()
    for k, shp in shapes.items():
        if k.endswith(".bias"):
            p[k] = torch.zeros(shp, dtype=torch.float32)
        elif kaiming:
            fan_in = 1
            for s_ in shp[1:]:
                fan_in *= s_
            p[k] = torch.randn(shp, generator=gen, dtype=torch.float32) * math.sqrt(2.0 / fan_in)
        else:
            p[k] = _xavier_normal(shp, gain, gen)
    return p


# --------------------------------------------------------------------------- #
# generator                                                                    #
# --------------------------------------------------------------------------- #
def positional_encoding(s, L):
    """gamma(s) = [s, sin(2^0 s), cos(2^0 s), ..., sin(2^(L-1) s), cos(2^(L-1) s)]
    (nerf-pytorch embedder convention, README.md:74)."""
    out = [s]
    for k in range(L):
        out.append(torch.sin(s * (2.0 ** k)))
        out.append(torch.cos(s * (2.0 ** k)))
    return torch.cat(out, dim=1)


def _act(x, kind, name, trace=None, masks=None):
    """ReLU / LeakyReLU(0.2) of a named pre-activation.  `trace` (dict) records the pre-activation under `name`.
    `masks` (dict name -> bool tensor) replaces the data-dependent branch `x > 0` by a given mask, which makes the
    network a smooth function of its parameters: the parity tests pass the masks the HIP forward actually took, so a
    pre-activation that rounds to the other side of 0 (|x| ~ 1e-7) does not show up as a gradient difference."""
    if trace is not None:
        trace[name] = x
    neg = 0.0 if kind == "relu" else LRELU
    if masks is not None and name in masks:
        return torch.where(masks[name], x, x * neg)
    return F.relu(x) if kind == "relu" else F.leaky_relu(x, LRELU)


def _pool(x, name, trace=None, masks=None):
    """F.max_pool2d(x, 2, 2); with `masks[name]` = int64 [N,C,H/2,W/2] window positions (0..3, row-major inside the
    2x2 window) the selection is taken from the mask instead of the data (see `_act`)."""
    N, C, H, W = x.shape
    Ho, Wo = H // 2, W // 2
    win = x[:, :, :Ho * 2, :Wo * 2].reshape(N, C, Ho, 2, Wo, 2).permute(0, 1, 2, 4, 3, 5).reshape(N, C, Ho, Wo, 4)
    if trace is not None:
        trace[name] = win.detach().argmax(4)
    if masks is not None and name in masks:
        return win.gather(4, masks[name].unsqueeze(-1)).squeeze(-1)
    return F.max_pool2d(x, 2, 2)


def _l1(a, b, name, masks=None, trace=None):
    """F.l1_loss(a, b) (mean); with `masks[name]` = sign(a - b) in {-1, 0, 1} the branch of |.| is given."""
    if trace is not None:
        trace[name] = torch.sign(a.detach() - b.detach())
    if masks is not None and name in masks:
        return (masks[name].to(a.dtype) * (a - b)).mean()
    return F.l1_loss(a, b)


def _hinge(x, sign, name, masks=None, trace=None):
    """mean(relu(1 + sign * x)); with `masks[name]` = (1 + sign * x > 0) the branch is given."""
    v = 1.0 + sign * x
    if trace is not None:
        trace[name] = v.detach() > 0
    if masks is not None and name in masks:
        return (masks[name].to(x.dtype) * v).mean()
    return F.relu(v).mean()


def state_mapping(p, state, spec, trace=None, masks=None):
    h = positional_encoding(state, spec.posenc_L)
    for i in range(spec.n_mlp):
        h = _act(F.linear(h, p[f"state_map.fc{i}.weight"], p[f"state_map.fc{i}.bias"]), "lrelu", f"state_map.fc{i}",
                 trace, masks)
    return h


def instance_norm(x):
    return F.instance_norm(x, eps=IN_EPS)


def mat_norm(p, prefix, x, prev_image, w):
    """MAT: IN(x) * (1 + gamma_img + gamma_st) + (beta_img + beta_st)   (SPEC.md D1;
    rebuttal.md:146-154: modulation parameters come from state AND image)."""
    seg = F.interpolate(prev_image, size=x.shape[2:], mode="nearest")
    actv = F.relu(F.conv2d(seg, p[prefix + ".mlp_shared.weight"], p[prefix + ".mlp_shared.bias"], padding=1))
    gamma = F.conv2d(actv, p[prefix + ".mlp_gamma.weight"], p[prefix + ".mlp_gamma.bias"], padding=1)
    beta = F.conv2d(actv, p[prefix + ".mlp_beta.weight"], p[prefix + ".mlp_beta.bias"], padding=1)
    st = F.linear(w, p[prefix + ".fc_state.weight"], p[prefix + ".fc_state.bias"])
    C = x.shape[1]
    g_st, b_st = st[:, :C, None, None], st[:, C:, None, None]
    return instance_norm(x) * (1.0 + gamma + g_st) + (beta + b_st)


def mat_resblock(p, b, x, prev_image, w, trace=None, masks=None):
    dx = F.conv2d(_act(mat_norm(p, f"blocks.{b}.norm_0", x, prev_image, w), "lrelu", f"blocks.{b}.norm_0", trace, masks),
                  p[f"blocks.{b}.conv_0.weight"], p[f"blocks.{b}.conv_0.bias"], padding=1)
    if trace is not None:
        trace[f"blocks.{b}.conv_0"] = dx
    dx = F.conv2d(_act(mat_norm(p, f"blocks.{b}.norm_1", dx, prev_image, w), "lrelu", f"blocks.{b}.norm_1", trace, masks),
                  p[f"blocks.{b}.conv_1.weight"], p[f"blocks.{b}.conv_1.bias"], padding=1)
    return x + dx


def generator_forward(p, prev_image, state, spec, trace=None, masks=None):
    """netG='s2p': (prev_image [N,3,H,W] in [-1,1], state [N,S]) -> image [N,3,H,W].
    `trace` / `masks`: see `_act` (pre-activations named stem, down{i}, blocks.{b}.norm_{j}, up{i}; the trace also holds
    the raw conv outputs `<name>.conv`, `blocks.{b}.conv_0`, `blocks.{b}.out` and the state code `w`)."""
    w = state_mapping(p, state, spec, trace, masks)
    x = F.conv2d(F.pad(prev_image, (3, 3, 3, 3), mode="reflect"), p["stem.weight"])
    if trace is not None:
        trace["w"] = w
        trace["stem.conv"] = x
    x = _act(instance_norm(x), "relu", "stem", trace, masks)
    for i in range(spec.n_down):
        x = F.conv2d(x, p[f"down{i}.weight"], stride=2, padding=1)
        if trace is not None:
            trace[f"down{i}.conv"] = x
        x = _act(instance_norm(x), "relu", f"down{i}", trace, masks)
    for b in range(spec.n_blocks):
        x = mat_resblock(p, b, x, prev_image, w, trace, masks)
        if trace is not None:
            trace[f"blocks.{b}.out"] = x
    for i in range(spec.n_down):
        x = F.conv_transpose2d(x, p[f"up{i}.weight"], stride=2, padding=1, output_padding=1)
        if trace is not None:
            trace[f"up{i}.conv"] = x
        x = _act(instance_norm(x), "relu", f"up{i}", trace, masks)
    x = F.conv2d(F.pad(x, (3, 3, 3, 3), mode="reflect"), p["out.weight"], p["out.bias"])
    return torch.tanh(x)


# --------------------------------------------------------------------------- #
# discriminator                                                                #
# --------------------------------------------------------------------------- #
def nlayer_discriminator(p, k, x, spec, trace=None, masks=None):
    """Pre-activations are named D{k}.model{n} (see `_act`)."""
    pre = f"discriminator_{k}"
    feats = []
    h = _act(F.conv2d(x, p[f"{pre}.model0.weight"], p[f"{pre}.model0.bias"], stride=2, padding=2), "lrelu",
             f"D{k}.model0", trace, masks)
    feats.append(h)
    for n in range(1, spec.n_layers_D):
        stride = 1 if n == spec.n_layers_D - 1 else 2
        h = F.conv2d(h, p[f"{pre}.model{n}.weight"], None, stride=stride, padding=2)
        h = _act(instance_norm(h), "lrelu", f"D{k}.model{n}", trace, masks)
        feats.append(h)
    n = spec.n_layers_D
    h = F.conv2d(h, p[f"{pre}.model{n}.weight"], p[f"{pre}.model{n}.bias"], stride=1, padding=2)
    feats.append(h)
    return feats


def multiscale_discriminator(p, x, spec, trace=None, masks=None):
    result = []
    for k in range(spec.num_D):
        result.append(nlayer_discriminator(p, k, x, spec, trace, masks))
        x = F.avg_pool2d(x, kernel_size=3, stride=2, padding=1, count_include_pad=False)
    return result


# --------------------------------------------------------------------------- #
# losses                                                                       #
# --------------------------------------------------------------------------- #
def vgg_features(p, x, trace=None, masks=None):
    """ReLU pre-activations are named vgg.<conv>, the pools vgg.pool{i} (see `_act`, `_pool`)."""
    feats = []
    h = x
    npool = 0
    for item in VGG_CFG:
        if item == "P":
            h = _pool(h, f"vgg.pool{npool}", trace, masks)
            npool += 1
            continue
        name = item[0]
        h = _act(F.conv2d(h, p[f"{name}.weight"], p[f"{name}.bias"], padding=1), "relu", f"vgg.{name}", trace, masks)
        if name in VGG_TAPS:
            feats.append(h)
    return feats


def hinge_d_loss(pred_fake, pred_real, masks=None, trace=None):
    """Mean over scales of mean(relu(1+D(fake))) / mean(relu(1-D(real))); last feature per scale."""
    lf = sum(_hinge(s[-1], 1.0, f"hinge.fake{k}", masks, trace) for k, s in enumerate(pred_fake)) / len(pred_fake)
    lr = sum(_hinge(s[-1], -1.0, f"hinge.real{k}", masks, trace) for k, s in enumerate(pred_real)) / len(pred_real)
    return lf, lr


def hinge_g_loss(pred_fake):
    return sum(-s[-1].mean() for s in pred_fake) / len(pred_fake)


def feat_match_loss(pred_fake, pred_real, lambda_feat, masks=None, trace=None):
    num_D = len(pred_fake)
    loss = 0.0
    for i in range(num_D):
        for j in range(len(pred_fake[i]) - 1):
            loss = loss + _l1(pred_fake[i][j], pred_real[i][j].detach(), f"l1.feat{i}.{j}", masks, trace) * lambda_feat / num_D
    return loss


def vgg_loss(pv, fake, real, masks=None, trace=None):
    """The product runs VGG once on cat([fake; real]) along the batch: masks of a VGG layer cover both halves."""
    n = fake.shape[0]
    f2 = vgg_features(pv, torch.cat([fake, real], 0), trace, masks)
    return sum(w * _l1(t[:n], t[n:].detach(), f"l1.vgg{k}", masks, trace) for k, (w, t) in enumerate(zip(VGG_WEIGHTS, f2)))


def discriminate(pd, prev_image, fake, real, spec, masks=None, trace=None):
    """One D call on cat([fake;real]) along batch, each concatenated with the conditioning
    previous image on channels (SPEC.md D7)."""
    x = torch.cat([torch.cat([prev_image, fake], 1), torch.cat([prev_image, real], 1)], 0)
    out = multiscale_discriminator(pd, x, spec, trace, masks)
    n = fake.shape[0]
    pf = [[t[:n] for t in s] for s in out]
    pr = [[t[n:] for t in s] for s in out]
    return pf, pr


def generator_losses(pg, pd, pv, prev_image, state, real, spec, masks=None, trace=None):
    """`masks`: optional branch masks for every data-dependent branch of the step (see `_act`, `_pool`, `_l1`);
    `trace` records the branches actually taken (`branch_masks` turns such a trace into masks)."""
    fake = generator_forward(pg, prev_image, state, spec, trace, masks)
    pf, pr = discriminate(pd, prev_image, fake, real, spec, masks, trace)
    losses = OrderedDict()
    losses["GAN"] = hinge_g_loss(pf)
    losses["GAN_Feat"] = feat_match_loss(pf, pr, spec.lambda_feat, masks, trace)
    if pv is not None:
        losses["VGG"] = vgg_loss(pv, fake, real, masks, trace) * spec.lambda_vgg
    losses["L1"] = _l1(fake, real, "l1.pix", masks, trace) * spec.lambda_l1
    return losses, fake


def branch_masks(trace):
    """Turn a trace into the mask dict that reproduces the same branches: pre-activations -> (x > 0); pool indices,
    L1 signs and hinge masks are recorded in mask form already.  Raw conv outputs etc. are dropped."""
    m = {}
    for k, v in trace.items():
        if k.startswith(("l1.", "hinge.", "vgg.pool")):
            m[k] = v
        elif k == "w" or k.endswith((".conv", ".conv_0", ".out")):
            continue
        else:
            m[k] = v.detach() > 0
    return m


def discriminator_losses(pg, pd, prev_image, state, real, spec, masks=None, fake=None, trace=None):
    """`fake`: optional pre-computed generator output (the parity tests pass the HIP generator's own output so that the
    D-step comparison isolates the discriminator path)."""
    if fake is None:
        with torch.no_grad():
            fake = generator_forward(pg, prev_image, state, spec)
    pf, pr = discriminate(pd, prev_image, fake.detach(), real, spec, masks, trace)
    lf, lr = hinge_d_loss(pf, pr, masks, trace)
    return OrderedDict(D_Fake=lf, D_real=lr)


def adam_step(p, g, m, v, step, lr, beta1, beta2, eps=1e-8):
    """torch.optim.Adam (no weight decay, no amsgrad) restated; returns new (p, m, v)."""
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    return p - (lr / bc1) * m / denom, m, v


def rollout(pg, image0, states, spec):
    """N-step autoregressive generation I_{t+1} = G(I_t, s_{t+1})  (README.md:30)."""
    frames = []
    img = image0
    with torch.no_grad():
        for t in range(states.shape[1]):
            img = generator_forward(pg, img, states[:, t], spec)
            frames.append(img)
    return torch.stack(frames, 1)
