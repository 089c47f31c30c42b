"""CPU: host-side logic of the product package -- plugin lookup, options, checkpoint naming, dataset, flat parameter
layout, C-ABI exports -- and that the product path refuses to run without a HIP device (no CPU fallback)."""
import os
import re
import sys

import numpy as np
import pytest
import torch

import s2p_oracle as O
from s2p_amd import _lib, ops
from s2p_amd.models import networks
from s2p_amd.options.test_options import TestOptions
from s2p_amd.options.train_options import TrainOptions

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_loads_and_exports_every_declared_symbol():
    L = _lib.lib()
    assert L.s2p_version() >= 109
    header = open(os.path.join(ROOT, "include", "s2p_hip.h")).read()
    declared = set(re.findall(r"\b(s2p_[a-z0-9_]+)\s*\(", header))
    declared -= {"s2p_conv_desc", "s2p_pack_job", "s2p_wgrad_job"}
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(L, name), "libs2p_hip.so does not export %s" % name
    assert declared == set(_lib.SIGNATURES.keys())
    # ... and nothing else: the library is built with -fvisibility=hidden, so no kernel stub or C++ helper leaks into the
    # dynamic symbol table (hygiene)
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib._SO], capture_output=True, text=True, check=True).stdout
    code = {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in ("T", "t", "W", "w")}
    assert code == declared, sorted(code ^ declared)


def test_python_product_does_not_select_a_library_from_the_environment():
    """`S2P_LIB` (a second build for A/B runs) is honoured by tools/uselib.py only; nothing under s2p_amd/ reads it."""
    import glob
    for f in glob.glob(os.path.join(ROOT, "s2p_amd", "**", "*.py"), recursive=True):
        assert "S2P_LIB" not in open(f).read(), f
    assert "S2P_LIB" in open(os.path.join(ROOT, "tools", "uselib.py")).read()


def test_product_library_reads_no_environment():
    """Diagnostics (timing ablations, A/B switches) live only in the -DS2P_DIAG_BUILD library: the product .so does not
    even import getenv, so a stray S2P_* variable in a training job cannot change a kernel."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--undefined-only", _lib._SO], capture_output=True, text=True).stdout
    assert "hipLaunchKernel" in out or "hipModuleLaunchKernel" in out or "__hipPushCallConfiguration" in out
    assert "getenv" not in out


def test_weight_gradient_workspace_plans_run_without_a_gpu():
    """Size queries are host logic (no launch): the K-split / tile-class plan of the padded-raster 4x4 weight gradient
    (csrc/wgrad_slabg.hip) for the PatchGAN layers of the train step -- classes x tiles x splits partial tiles of 64 x T x 64
    floats + the bias partials -- and the queries' refusal of what no kernel takes."""
    import ctypes
    from s2p_amd import ops
    L = _lib.lib()

    def need(cin, cout, k, s, p, N, H, W):
        geom = ops.ConvGeom(cin, cout, k, s, p)
        d = geom.desc(torch.bfloat16, N, H, W, cin, cin, cout)
        return int(L.s2p_conv2d_wgrad_workspace(ctypes.byref(d), cin, cout))

    # 256 -> 512, 4x4 stride 1 pad 2 on 12x12 (N 128): 2 classes (8 taps) x 32 tiles x 8 splits; at least that many partial tiles
    assert need(256, 512, 4, 1, 2, 128, 12, 12) >= (2 * 32 * 8 * 64 * 8 * 64 + 8 * 8 * 64) * 4
    # 64 -> 128, 4x4 stride 2 on 43x43: 4 parity classes (4 taps) x 2 tiles x 63 splits (1 058 raster blocks of 64 positions in
    # runs of 17) of 64 x 4 x 64 floats + 2 x 63 x 64 bias partials: exactly this many bytes
    assert need(64, 128, 4, 2, 2, 128, 43, 43) == (4 * 2 * 63 * 64 * 4 * 64 + 2 * 63 * 64) * 4
    # a layer neither weight-gradient kernel family splits deterministically asks for nothing
    assert need(64, 128, 3, 2, 1, 4, 20, 20) > 0            # implicit GEMM: K-split partial tiles
    d = ops.ConvGeom(64, 128, 4, 2, 2).desc(torch.float32, 2, 9, 9, 64, 64, 128)
    assert int(L.s2p_conv2d_wgrad_workspace(ctypes.byref(d), 64, 128)) == 0     # fp32 path: atomics, no scratch


def test_fused_norm_dispatch_is_host_logic():
    """`s2p_conv2d_mat_is_fused` is a plan-mode dry run of the conv dispatcher (no launch): which conv + InstanceNorm pairs of the train
    step are ONE launch is decided on the host and can be pinned without a GPU -- the ResBlk convs with their MAT norms forward and
    backward (conv_plane.hip), the PatchGAN 4x4 stride-1 layers forward and backward and the stride-2 layers forward on planes of up to
    192 produced pixels (conv_planeg.hip); the 43x43 -> 22x22 layer, the stride-2 dgrads (sub-pixel phases), row-band planes and the
    64x64 maps of the 256x256 rollout are two launches."""
    import ctypes
    from s2p_amd import ops
    L = _lib.lib()

    def fused(cin, cout, k, s, p, N, H, W, dgrad, gb):
        d = ops.ConvGeom(cin, cout, k, s, p).desc(torch.bfloat16, N, H, W, cin, cin, cout)
        return int(L.s2p_conv2d_mat_is_fused(ctypes.byref(d), dgrad, gb))

    one = [(256, 256, 3, 1, 1, 64, 21, 21, 0, 1), (256, 256, 3, 1, 1, 64, 21, 21, 1, 1), (256, 512, 4, 1, 2, 64, 12, 12, 0, 0),
           (256, 512, 4, 1, 2, 64, 12, 12, 1, 0), (256, 512, 4, 1, 2, 64, 7, 7, 0, 0), (128, 256, 4, 2, 2, 64, 22, 22, 0, 0),
           (64, 128, 4, 2, 2, 64, 22, 22, 0, 0)]
    two = [(64, 128, 4, 2, 2, 64, 43, 43, 0, 0), (128, 256, 4, 2, 2, 64, 22, 22, 1, 0), (64, 64, 3, 1, 1, 64, 84, 84, 0, 0),
           (256, 256, 3, 1, 1, 16, 64, 64, 0, 1)]
    for a in one:
        assert fused(*a) == 1, a
    for a in two:
        assert fused(*a) == 0, a


def test_struct_layouts_match_header():
    import ctypes
    assert ctypes.sizeof(_lib.ConvDesc) == 20 * 4
    assert ctypes.sizeof(_lib.PackJob) == 56      # 3 pointers + 7 int32, padded to 8-byte alignment (same in C)


def test_netG_s2p_plugin_lookup_and_state_dict_contract():
    cls = networks.find_network_using_name("s2p", "generator")
    assert cls.__name__ == "S2PGenerator"
    assert networks.find_network_using_name("multiscale", "discriminator").__name__ == "MultiscaleDiscriminator"
    with pytest.raises(ValueError):
        networks.find_network_using_name("nope", "generator")
    opt = TrainOptions().parse(["--env_type", "walker", "--gpu_ids", "0"], quiet=True)
    assert opt.state_dim == 24 and opt.netG == "s2p" and opt.gpu_ids == [0]
    netG = networks.define_G(opt)
    shapes = O.generator_param_shapes(O.Spec(state_dim=24))
    sd = netG.state_dict()
    assert sorted(sd.keys()) == sorted(shapes.keys())
    for k, shp in shapes.items():
        assert tuple(sd[k].shape) == tuple(shp), k
    netD = networks.define_D(opt)
    dshapes = O.discriminator_param_shapes(O.Spec())
    assert {k: tuple(v.shape) for k, v in netD.state_dict().items()} == {k: tuple(v) for k, v in dshapes.items()}
    # SPADE-style init: xavier-normal(0.02) weights, zero biases
    assert float(sd["out.bias"].abs().max()) == 0.0 and 0 < float(sd["stem.weight"].std()) < 0.01


def test_readme_command_lines_parse():
    o = TestOptions().parse("--env_type=cheetah --dataroot=./datasets --netG=s2p --start_idx=0 --seq_len=5 --gpu_ids=0".split(), quiet=True)
    assert (o.env_type, o.netG, o.start_idx, o.seq_len, o.gpu_ids, o.state_dim) == ("cheetah", "s2p", 0, 5, [0], 17)
    t = TrainOptions().parse("--dataroot=./datasets/cheetah.hdf5 --env_type=cheetah --netG=s2p --batchSize=16 --gpu_ids=0".split(), quiet=True)
    assert t.batchSize == 16 and t.isTrain and t.lambda_feat == 10.0 and t.num_D == 2 and t.n_layers_D == 4


def test_no_cpu_fallback():
    from s2p_amd.models.pix2pix_model import Pix2PixModel
    o = TestOptions().parse(["--gpu_ids", "-1", "--random_init"], quiet=True)
    with pytest.raises(RuntimeError, match="HIP device"):
        Pix2PixModel(o)
    if not torch.cuda.is_available():
        o2 = TestOptions().parse(["--gpu_ids", "0", "--random_init"], quiet=True)
        with pytest.raises(RuntimeError, match="HIP"):
            Pix2PixModel(o2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.avgpool_fwd(torch.zeros(1, 4, 4, 4))
    opt = TrainOptions().parse(["--gpu_ids", "0"], quiet=True)
    netG = networks.define_G(opt)
    with pytest.raises(RuntimeError, match="finalize|HIP"):
        netG(torch.zeros(1, 3, 84, 84), torch.zeros(1, 17))
    with pytest.raises(RuntimeError, match="HIP device only"):
        netG.finalize(torch.device("cpu"), torch.float32)
    # the product package never imports the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "s2p_amd")):
        for f in files:
            if f.endswith(".py"):
                assert "s2p_oracle" not in open(os.path.join(dirpath, f)).read(), f


def test_flat_param_store_layout_on_cpu():
    """Flat master/grad buffers: parameters become channels-last views; fused packs are contiguous runs."""
    opt = TrainOptions().parse(["--gpu_ids", "0"], quiet=True)
    netG = networks.define_G(opt)
    before = {k: v.detach().clone() for k, v in netG.state_dict().items()}
    netG._declare_packs(torch.bfloat16)
    st = netG.store
    st.finalize(torch.device("cpu"))            # layout only: packing itself is a HIP launch
    assert st.numel == sum(v.numel() for v in before.values())
    for k, v in netG.state_dict().items():
        assert torch.equal(v, before[k]), k
    w = netG.blocks[0].conv_0.weight
    assert w.shape == (256, 256, 3, 3) and w.stride() == (9 * 256, 1, 3 * 256, 256)      # physically [Co][kh][kw][Ci]
    assert w.grad is not None and w.grad.data_ptr() >= st.grad.data_ptr()
    gb = netG.lay["gb"].pk
    assert gb.groups == 12 and gb.R == 512 and gb.fwd.shape == (12, 512, 9, 128) and gb.bwd.shape == (12, 128, 9, 512)
    assert gb.gw.numel() == 12 * 512 * 9 * 128 and gb.bias.numel() == 12 * 512
    sh = netG.lay["shared"].pk
    assert sh.fwd.shape == (1, 12 * 128, 9, 8) and sh.bwd is None
    up = netG.lay["up0"].pk
    assert up.kind == "convT" and up.w_fwd is up.bwd and up.w_fwd.shape == (1, 128, 9, 256)
    # writing through the flat buffer is visible through the parameter views (what the fused Adam relies on)
    st.master.add_(1.0)
    assert torch.allclose(netG.out.bias, before["out.bias"] + 1.0)
    ck = netG.export_state_dict()
    assert all(v.is_contiguous() for v in ck.values())
    # the checkpointed layout signature names every entry: swapping two same-shaped parameters changes it, and an
    # optimizer state without a signature is refused instead of being applied on numel alone
    names = {id(p): n for n, p in netG.named_parameters()}
    sig = st.layout_signature(names)
    a, b = netG.blocks[0].conv_0.weight, netG.blocks[1].conv_0.weight
    swapped = dict(names); swapped[id(a)], swapped[id(b)] = names[id(b)], names[id(a)]
    assert st.layout_signature(names) == sig and st.layout_signature(swapped) != sig and sig.startswith("v2:") and len(sig) == 19
    with pytest.raises(RuntimeError, match="no flat-layout signature"):
        st.load_optimizer_state(dict(step=1, m=torch.zeros(st.numel), v=torch.zeros(st.numel)))
    # the signature is versioned; a state that carries the round-3 hash of an IDENTICAL layout (offset, numel, shape
    # only), or the round-4 hash without the prefix, still loads; an unsigned one loads only with the explicit opt-in that
    # FlatAdam.load_state_dict / `--allow_unsigned_optimizer_state` pass down; a foreign signature is refused
    st.param_names = names
    z = dict(step=3, m=torch.zeros(st.numel), v=torch.zeros(st.numel))
    st.load_optimizer_state(dict(z, layout=st.legacy_layout_signature()))
    st.load_optimizer_state(dict(z, layout=sig[3:]))
    st.load_optimizer_state(dict(z, layout=sig))
    st.load_optimizer_state(dict(z), allow_unsigned=True)
    assert st.step == 3
    with pytest.raises(RuntimeError, match="different flat parameter layout"):
        st.load_optimizer_state(dict(z, layout="v2:0123456789abcdef"))
    popt = TrainOptions().parse(["--gpu_ids", "0", "--allow_unsigned_optimizer_state"], quiet=True)
    assert popt.allow_unsigned_optimizer_state and not opt.allow_unsigned_optimizer_state
    # data-parallel exchange buckets: they partition the early-complete tail of the flat gradient, the first (the one the backward
    # completes first) ends at the end of the buffer, and at least 2/3 of the tail is in the buckets before the last one
    bk = netG.early_buckets()
    assert len(bk) == 3 and all(len(r) == 3 for r in bk)
    spans = sorted((o, o + n) for r in bk for o, n in r)
    assert spans[0][0] == netG.early_grad_offset and spans[-1][1] == st.numel
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])) and sum(b - a for a, b in spans) == st.numel - netG.early_grad_offset
    assert max(o + n for o, n in bk[0]) == st.numel
    assert sum(n for r in bk[:-1] for _, n in r) >= 0.66 * (st.numel - netG.early_grad_offset)


def test_dataset_and_checkpoint_naming(tmp_path):
    from s2p_amd.data import S2PDataset, images_to_tensor, tensor_to_images
    opt = TrainOptions().parse(["--dataroot", os.path.join(ROOT, "datasets"), "--env_type", "cheetah", "--gpu_ids", "0"], quiet=True)
    ds = S2PDataset(opt)
    assert len(ds) == 11
    it = ds[3]
    assert it["prev_image"].shape == (3, 84, 84) and it["state"].shape == (17,) and it["image"].shape == (3, 84, 84)
    assert float(it["prev_image"].min()) >= -1 and float(it["prev_image"].max()) <= 1
    frames, states = ds.sequence(0, 5)
    assert frames.shape == (6, 3, 84, 84) and states.shape == (6, 17)
    with pytest.raises(IndexError):
        ds.sequence(8, 5)
    u8 = (np.random.default_rng(0).integers(0, 256, (2, 84, 84, 3))).astype(np.uint8)
    assert np.array_equal(tensor_to_images(images_to_tensor(u8)), u8)          # uint8 round trip is exact
    # episode boundaries are never crossed
    z = dict(np.load(os.path.join(ROOT, "datasets", "cheetah.npz")))
    z["timeouts"][4] = True
    np.savez(os.path.join(tmp_path, "cheetah.npz"), **z)
    opt.dataroot = str(tmp_path)
    assert 4 not in S2PDataset(opt).index
    from s2p_amd.models.pix2pix_model import Pix2PixModel
    m = Pix2PixModel.__new__(Pix2PixModel)
    m.opt = opt
    opt.checkpoints_dir = "./checkpoints"
    assert m.ckpt_path(30) == "./checkpoints/cheetah_30.pth"                    # README.md:19-26


def test_conv_geometry_matches_torch():
    import torch.nn.functional as F
    for (cin, cout, k, s, p, tr, op, H) in [(3, 8, 7, 1, 3, False, 0, 20), (8, 8, 3, 2, 1, False, 0, 21), (8, 4, 3, 2, 1, True, 1, 11),
                                             (6, 8, 4, 2, 2, False, 0, 84), (8, 8, 4, 1, 2, False, 0, 12)]:
        g = ops.ConvGeom(cin, cout, k, s, p, transposed=tr, output_padding=op)
        x = torch.zeros(1, cin, H, H)
        y = F.conv_transpose2d(x, torch.zeros(cin, cout, k, k), stride=s, padding=p, output_padding=op) if tr else \
            F.conv2d(x, torch.zeros(cout, cin, k, k), stride=s, padding=p)
        assert g.out_hw(H, H) == tuple(y.shape[2:])


def _audit():
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import isa_audit
    if not os.path.exists(isa_audit.LLVM + "/llvm-objdump"):
        pytest.skip("llvm-objdump not available")
    a = isa_audit.audit(_lib._SO)
    assert len(a) > 60, "disassembly found too few kernels"
    return a, isa_audit.demangle(list(a))


def test_no_packed_fp32_instructions():
    """ISA-level guard for the co-residency wrong-result hazard, no GPU needed.  Round 4 traced it to ONE
    instruction class: a PACKED fp32 VALU instruction with an op_sel operand swizzle (`v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0]`,
    `v_pk_fma_f32 ... op_sel_hi:[1,1,0]` -- the forms hipcc emits when it SLP-vectorises scalar fp32 arithmetic) returns wrong
    results in lanes 48..63 while its wave shares a SIMD with the slab weight-gradient kernel or the LDS-DMA conv kernel; never when
    alone (tests/tools/repro_valu_probe.py: 142 224 of 786 432 lanes wrong, all in 48..63; the same chain without op_sel, and every
    other instruction form tried, 0).  That is what the state path's linear kernels (rounds 2-3: "merged LDS reads" -- the merged
    reads only put the operands in adjacent registers, which let the SLP vectoriser pack the FMAs) and the SSIM kernel (round 3:
    the LDS tiles were innocent) suffered from.  The library is therefore built WITHOUT packed fp32 instructions
    (`-target-feature -packed-fp32-ops`, csrc/build.sh), and this test disassembles every gfx950 kernel of libs2p_hip.so and requires
      * no v_pk_*_f32 instruction at all, and
      * no `op_sel` operand modifier on any instruction (the other packed forms were clean in the probe WITHOUT a swizzle;
        a swizzled one would have to be probed first)."""
    import re
    a, names = _audit()
    bad = {names[k]: dict(v["packed"]) for k, v in a.items() if v["packed"]}
    assert not bad, bad


def test_lds_free_kernels_stay_lds_free():
    """The state path's linear kernels and the PSNR / SSIM kernel were rewritten without LDS while the hazard above was still
    believed to be an LDS effect (rounds 3-4).  That belief was wrong, but the LDS-free forms are the faster ones (fp32 MFMAs
    fed from global memory; a register window filter) and stay: keep them honest."""
    import re
    a, names = _audit()
    lin = [k for k in a if re.search(r"lin_(fwd|wgrad)_kernel", names[k])]
    assert len(lin) == 2, [names[k] for k in lin]
    met = [k for k in a if re.search(r"image_metrics_kernel", names[k])]
    assert len(met) == 1, [names[k] for k in met]
    for k in lin + met:
        assert not a[k]["ds"] and not a[k]["lds_dma"], (names[k], dict(a[k]["ds"]))


def test_rccl_sum_reduction_is_free_of_the_hazard_class():
    """The data-parallel step runs RCCL's reduction kernels on the communication stream WHILE `wgrad_slab_kernel` / `conv_dma_kernel`
    execute: the two aggressors of the co-residency hazard.  libs2p_hip.so is clean of the victim instruction
    class by construction; RCCL is third-party code, so its gfx950 code object is audited here (no GPU: the code object is extracted
    from the librccl.so torch loads with clang-offload-bundler, tests/tools/rccl_audit.py).  Finding (profiles/round5_rccl_isa_audit.txt):
    RCCL's device code is per-(reduction, type) device functions behind a few generic kernels; the class -- `v_pk_fma_f32 ... op_sel`
    -- occurs ONLY in the FuncPreMulSum<float> functions (ncclAvg / premul-sum).  The product reduces with SUM (the 1/world factor
    is folded into Adam's grad_scale): FuncSum<float> is scalar `v_add_f32` in every ring function, and `v_pk_add_f32` WITHOUT a
    swizzle in the tree functions (the form that was clean in round 4's probe beside both aggressors).  Hence:
      * no instruction of the class in any function a SUM / fp32 or MAX / f64 all-reduce can reach, and
      * the product never asks for AVG / PREMUL_SUM (s2p_amd/ is grepped), which would select the functions that DO carry it."""
    import re
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import rccl_audit
    so = rccl_audit.find_librccl()
    if so is None or not os.path.exists(rccl_audit.LLVM + "/clang-offload-bundler"):
        pytest.skip("no librccl.so / clang-offload-bundler in this environment")
    rep = rccl_audit.product_report(so)
    for key in rccl_audit.PRODUCT_REDUCTIONS:
        r = rep["reductions"][key]
        assert r["functions"] >= 20, (key, r["functions"])            # the symbol pattern still matches this RCCL build
        assert not r["hazard"], (key, r["hazard"], r["functions_with_packed"])
    ring = [n for n in rep["reductions"]["sum_f32"]["functions_with_packed"] if "runRing" in n]
    assert not ring, ring                                               # ring SUM: scalar fp32 adds only
    # the product's collectives: SUM (gradients), MAX (bench timing), broadcast, all_gather -- never AVG / PREMUL_SUM
    for dirpath, _, files in os.walk(os.path.join(ROOT, "s2p_amd")):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"ReduceOp\.(AVG|PREMUL_SUM)|_make_nccl_premul_sum", src), os.path.join(dirpath, fn)


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("s2p_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_resblk_aggregate_counts_generator_launches_only():
    """VGG conv3_x has the ResBlk conv's shape (64, 21, 21, 256, 256, 3, 1, 1, 0); the north_star aggregate must not
    pick it up.  Records carry the network tag of their layer (ops._Prof <- ConvGeom.net); the aggregate filters on it."""
    b = _load_bench()
    shp = (64, 21, 21, 256, 256, 3, 1, 1, 0)
    fl = 2.0 * 64 * 21 * 21 * 256 * 256 * 9
    recs = [dict(kind="fwd", shape=shp, flops=fl, ms=0.040, net="G") for _ in range(24)]
    recs += [dict(kind="dgrad", shape=shp, flops=fl, ms=0.045, net="G") for _ in range(12)]
    with_vgg = recs + [dict(kind="fwd", shape=shp, flops=fl, ms=0.032, net="VGG") for _ in range(6)] + \
        [dict(kind="dgrad", shape=shp, flops=fl, ms=0.030, net="VGG") for _ in range(3)]
    a0, a1 = b.spade_resblk_aggregate(recs, 64, 84), b.spade_resblk_aggregate(with_vgg, 64, 84)
    assert a0 is not None and a0["gflop"] == a1["gflop"] and a0["ms"] == a1["ms"] and a0["frac"] == a1["frac"]
    assert abs(a0["gflop"] - (12 + 12) * fl / 1e9) < 0.1           # forward launches count half (the step runs G forward twice)
    assert "generator launches only" in a0["includes"]
    from s2p_amd.ops import ConvGeom
    assert ConvGeom(3, 8, 3).net == "" and ConvGeom(3, 8, 3, net="D").net == "D"


def test_bench_hbm_rows_from_the_committed_profiles():
    """`roofline.hbm_rows` is built from the latest profiles/round*_pmc_traffic.json + the serial kernel summary of
    the same round; it refuses files measured on another library version (bench.py then reports null)."""
    import glob, json
    b = _load_bench()
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_traffic.json")))
    assert cands, "no PMC evidence committed"
    ver = json.load(open(cands[-1]))["_meta"]["s2p_version"]
    h = b.hbm_rows(64, 84, ver)
    assert h["peak"] == 8000.0 and h["unit"] == "GB/s" and len(h["rows"]) == 3
    for r in h["rows"]:
        assert 0.0 < r["frac"] < 1.0 and r["mfma_busy_share"] < 0.02 and r["achieved_gbs"] == pytest.approx(
            (r["hbm_read_mb"] + r["hbm_write_mb"]) / r["avg_us"] * 1e3, rel=1e-2)
    assert 5.0 < h["whole_step"]["gb_per_step"] < 60.0 and h["whole_step"]["ms_at_peak"] == pytest.approx(h["whole_step"]["gb_per_step"] / 8.0, rel=1e-2)
    assert all(a["read_amplification"] >= 0.9 for a in h["read_amplification"])
    with pytest.raises(RuntimeError, match="library version"):
        b.hbm_rows(64, 84, ver + 1000)
