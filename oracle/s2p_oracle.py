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


def state_mapping(p, state, spec):
    h = positional_encoding(state, spec.posenc_L)
    for i in range(spec.n_mlp):
        h = F.leaky_relu(F.linear(h, p[f"state_map.fc{i}.weight"], p[f"state_map.fc{i}.bias"]), LRELU)
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


def mat_resblock(p, b, x, prev_image, w):
    dx = F.conv2d(F.leaky_relu(mat_norm(p, f"blocks.{b}.norm_0", x, prev_image, w), LRELU),
                  p[f"blocks.{b}.conv_0.weight"], p[f"blocks.{b}.conv_0.bias"], padding=1)
    dx = F.conv2d(F.leaky_relu(mat_norm(p, f"blocks.{b}.norm_1", dx, prev_image, w), LRELU),
                  p[f"blocks.{b}.conv_1.weight"], p[f"blocks.{b}.conv_1.bias"], padding=1)
    return x + dx


def generator_forward(p, prev_image, state, spec):
    """netG='s2p': (prev_image [N,3,H,W] in [-1,1], state [N,S]) -> image [N,3,H,W]."""
    w = state_mapping(p, state, spec)
    x = F.conv2d(F.pad(prev_image, (3, 3, 3, 3), mode="reflect"), p["stem.weight"])
    x = F.relu(instance_norm(x))
    for i in range(spec.n_down):
        x = F.relu(instance_norm(F.conv2d(x, p[f"down{i}.weight"], stride=2, padding=1)))
    for b in range(spec.n_blocks):
        x = mat_resblock(p, b, x, prev_image, w)
    for i in range(spec.n_down):
        x = F.conv_transpose2d(x, p[f"up{i}.weight"], stride=2, padding=1, output_padding=1)
        x = F.relu(instance_norm(x))
    x = F.conv2d(F.pad(x, (3, 3, 3, 3), mode="reflect"), p["out.weight"], p["out.bias"])
    return torch.tanh(x)


# --------------------------------------------------------------------------- #
# discriminator                                                                #
# --------------------------------------------------------------------------- #
def nlayer_discriminator(p, k, x, spec):
    pre = f"discriminator_{k}"
    feats = []
    h = F.leaky_relu(F.conv2d(x, p[f"{pre}.model0.weight"], p[f"{pre}.model0.bias"], stride=2, padding=2), LRELU)
    feats.append(h)
    for n in range(1, spec.n_layers_D):
        stride = 1 if n == spec.n_layers_D - 1 else 2
        h = F.conv2d(h, p[f"{pre}.model{n}.weight"], None, stride=stride, padding=2)
        h = F.leaky_relu(instance_norm(h), LRELU)
        feats.append(h)
    n = spec.n_layers_D
    h = F.conv2d(h, p[f"{pre}.model{n}.weight"], p[f"{pre}.model{n}.bias"], stride=1, padding=2)
    feats.append(h)
    return feats


def multiscale_discriminator(p, x, spec):
    result = []
    for k in range(spec.num_D):
        result.append(nlayer_discriminator(p, k, x, spec))
        x = F.avg_pool2d(x, kernel_size=3, stride=2, padding=1, count_include_pad=False)
    return result


# --------------------------------------------------------------------------- #
# losses                                                                       #
# --------------------------------------------------------------------------- #
def vgg_features(p, x):
    feats = []
    h = x
    for item in VGG_CFG:
        if item == "P":
            h = F.max_pool2d(h, 2, 2)
            continue
        name = item[0]
        h = F.relu(F.conv2d(h, p[f"{name}.weight"], p[f"{name}.bias"], padding=1))
        if name in VGG_TAPS:
            feats.append(h)
    return feats


def hinge_d_loss(pred_fake, pred_real):
    """Mean over scales of mean(relu(1+D(fake))) / mean(relu(1-D(real))); last feature per scale."""
    lf = sum(F.relu(1.0 + s[-1]).mean() for s in pred_fake) / len(pred_fake)
    lr = sum(F.relu(1.0 - s[-1]).mean() for s in pred_real) / len(pred_real)
    return lf, lr


def hinge_g_loss(pred_fake):
    return sum(-s[-1].mean() for s in pred_fake) / len(pred_fake)


def feat_match_loss(pred_fake, pred_real, lambda_feat):
    num_D = len(pred_fake)
    loss = 0.0
    for i in range(num_D):
        for j in range(len(pred_fake[i]) - 1):
            loss = loss + F.l1_loss(pred_fake[i][j], pred_real[i][j].detach()) * lambda_feat / num_D
    return loss


def vgg_loss(pv, fake, real):
    ff, fr = vgg_features(pv, fake), vgg_features(pv, real)
    return sum(w * F.l1_loss(a, b.detach()) for w, a, b in zip(VGG_WEIGHTS, ff, fr))


def discriminate(pd, prev_image, fake, real, spec):
    """One D call on cat([fake;real]) along batch, each concatenated with the conditioning
    previous image on channels (SPEC.md D7)."""
    x = torch.cat([torch.cat([prev_image, fake], 1), torch.cat([prev_image, real], 1)], 0)
    out = multiscale_discriminator(pd, x, spec)
    n = fake.shape[0]
    pf = [[t[:n] for t in s] for s in out]
    pr = [[t[n:] for t in s] for s in out]
    return pf, pr


def generator_losses(pg, pd, pv, prev_image, state, real, spec):
    fake = generator_forward(pg, prev_image, state, spec)
    pf, pr = discriminate(pd, prev_image, fake, real, spec)
    losses = OrderedDict()
    losses["GAN"] = hinge_g_loss(pf)
    losses["GAN_Feat"] = feat_match_loss(pf, pr, spec.lambda_feat)
    if pv is not None:
        losses["VGG"] = vgg_loss(pv, fake, real) * spec.lambda_vgg
    losses["L1"] = F.l1_loss(fake, real) * spec.lambda_l1
    return losses, fake


def discriminator_losses(pg, pd, prev_image, state, real, spec):
    with torch.no_grad():
        fake = generator_forward(pg, prev_image, state, spec)
    pf, pr = discriminate(pd, prev_image, fake.detach(), real, spec)
    lf, lr = hinge_d_loss(pf, pr)
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
