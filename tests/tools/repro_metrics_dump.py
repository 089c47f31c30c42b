"""Where does round 3's LDS-staged SSIM kernel lose data beside the slab weight-gradient kernel?  
The diagnostics build keeps that kernel (`s2p_diag_image_metrics_lds`) with a dump of its LDS tiles: `pa`, `pb` copied out
after the first barrier, `hm` after the second, plus HW_ID / LDS_ALLOC / XCC_ID of every workgroup.  ONE quiet run and ONE run
beside the aggressor are diffed: which array differs first, which rows / columns / lanes, and what the wrong words hold.
    S2P_LIB=.../libs2p_hip_diag.so python tests/tools/repro_metrics_dump.py"""
import ctypes, os, sys, io, contextlib, collections
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import _lib
from s2p_amd.models.networks.layers import ConvLayer
from s2p_amd.options.train_options import TrainOptions
from s2p_amd.models.pix2pix_model import Pix2PixModel

opt = TrainOptions().parse(["--env_type", "cheetah", "--batchSize", "64", "--precision", "bf16", "--gpu_ids", "0", "--checkpoints_dir", "/tmp/ck_md"], quiet=True)
with contextlib.redirect_stdout(io.StringIO()):
    m = Pix2PixModel(opt)
L = m.netG.lay
g = torch.Generator().manual_seed(0)
bf = torch.bfloat16
a21 = torch.randn(64, 21, 21, 256, generator=g).to(bf).cuda(); b21 = torch.randn(64, 21, 21, 256, generator=g).to(bf).cuda()
a84 = torch.randn(64, 84, 84, 64, generator=g).to(bf).cuda()
N, C, H, W = 8, 3, 84, 84
img_a = (torch.rand(N, C, H, W, generator=g).cuda() * 2 - 1).contiguous()
img_b = (img_a + 0.1 * torch.randn(N, C, H, W, generator=g).cuda()).clamp(-1, 1).contiguous()
lib = ctypes.CDLL(_lib._SO)
P = ctypes.c_void_p
lib.s2p_diag_image_metrics_lds.argtypes = [P, P] + [ctypes.c_int] * 4 + [ctypes.c_float] + [P] * 5
DF = lib.s2p_diag_image_metrics_dump_floats()
PATCH, TILE = 26, 16
tiles = (H - 10 + TILE - 1) // TILE
nwg = tiles * tiles * N * C
side = torch.cuda.Stream()


def run(stream, dump=True):
    acc = torch.zeros(2, N, device="cuda")
    d = torch.full((nwg, DF), float("nan"), device="cuda") if dump else None
    hw = torch.zeros(nwg, 4, dtype=torch.int32, device="cuda")
    rc = lib.s2p_diag_image_metrics_lds(img_a.data_ptr(), img_b.data_ptr(), N, C, H, W, 2.0, acc[0].data_ptr(), acc[1].data_ptr(),
                                        d.data_ptr() if dump else None, hw.data_ptr(), stream.cuda_stream)
    assert rc == 0
    return acc, d, hw


def split(d):
    n1 = PATCH * (PATCH + 1)
    return (d[:, :n1].view(-1, PATCH, PATCH + 1), d[:, n1:2 * n1].view(-1, PATCH, PATCH + 1),
            d[:, 2 * n1:].view(-1, 5, PATCH, TILE + 1))


def decode_hw(h):
    hw_id, lds, xcc = int(h[0]) & 0xffffffff, int(h[1]) & 0xffffffff, int(h[2]) & 0xffffffff
    return dict(wave=hw_id & 0xf, simd=(hw_id >> 4) & 3, cu=(hw_id >> 8) & 0xf, sh=(hw_id >> 12) & 1, se=(hw_id >> 13) & 7,
                lds_base=lds & 0xff, lds_size=(lds >> 12) & 0x1ff, lds_raw=hex(lds), xcc=xcc & 0xf, pa_addr=int(h[3]))


torch.cuda.synchronize()
qa, qd, qh = run(torch.cuda.current_stream())
torch.cuda.synchronize()
qa2, qd2, _ = run(torch.cuda.current_stream())
torch.cuda.synchronize()
print("quiet vs quiet: sums equal %s, tiles equal %s" % (torch.allclose(qa, qa2, rtol=1e-6), torch.equal(torch.nan_to_num(qd), torch.nan_to_num(qd2))))
print("quiet LDS_ALLOC census:", collections.Counter(decode_hw(h)["lds_raw"] for h in qh.cpu()).most_common(6))

aggressors = {
    "slab weight gradient": lambda: ConvLayer.wgrad_many([(L["b0c0"], a21, b21), (L["b0c1"], a21, b21)]),
    "LDS-DMA conv (down0 forward)": lambda: L["down0"].fwd(a84),
}
for aname, afn in aggressors.items():
    for _ in range(6):
        afn()
    with torch.cuda.stream(side):
        ba, bd, bh = run(side)
    torch.cuda.synchronize()
    rel = ((ba - qa).abs() / qa.abs()).max(dim=1).values.tolist()
    print("\n=== beside %s: max rel diff of the sums: sq %.3e ssim %.3e" % (aname, rel[0], rel[1]))
    names = ("pa", "pb", "hm")
    for nm, q, b in zip(names, split(qd), split(bd)):
        q = torch.nan_to_num(q, nan=-7.0); b = torch.nan_to_num(b, nan=-7.0)
        diff = (q != b)
        nbad = int(diff.sum())
        wgs = diff.flatten(1).any(1).nonzero().flatten().tolist()
        print("  %s: %d words differ in %d of %d workgroups" % (nm, nbad, len(wgs), nwg))
        if not nbad:
            continue
        idx = diff.nonzero()
        vals = b[diff]
        print("     wrong words that are exactly 0: %d; NaN-filled (never written to the dump): %d" % (int((vals == 0).sum()), int((vals == -7.0).sum())))
        cols = collections.Counter(idx[:, -1].tolist()); rows = collections.Counter(idx[:, -2].tolist())
        print("     by last index (column):", sorted(cols.items()))
        print("     by row:", sorted(rows.items()))
        if nm == "hm":
            print("     by moment:", sorted(collections.Counter(idx[:, 1].tolist()).items()))
        for w in wgs[:6]:
            sel = idx[idx[:, 0] == w]
            ex = sel[0].tolist()
            print("     wg %d %s: %d words, first at %s quiet %.6g beside %.6g" % (w, decode_hw(bh[w].cpu()), len(sel), ex[1:], float(q[tuple(ex)]), float(b[tuple(ex)])))
    bad_wgs = set()
    for q, b in zip(split(qd), split(bd)):
        bad_wgs |= set((torch.nan_to_num(q, nan=-7.0) != torch.nan_to_num(b, nan=-7.0)).flatten(1).any(1).nonzero().flatten().tolist())
    allhw = [decode_hw(h) for h in bh.cpu()]
    print("  LDS_ALLOC of the bad workgroups:", collections.Counter(allhw[w]["lds_raw"] for w in bad_wgs).most_common(8))
    print("  LDS_ALLOC of all workgroups   :", collections.Counter(h["lds_raw"] for h in allhw).most_common(8))
