"""bench.py -- BASELINE.json metric: G+D train-step images/sec at 84x84, bs=64/GPU, on 1/2/4/8 MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = run_generator_one_step + run_discriminator_one_step (hinge GAN + feature matching + VGG19 perceptual +
pixel L1, both Adam updates) on one synthetic batch of 64 (prev_image, state, image) triples per GPU, bf16 compute /
fp32 accumulate, inputs resident in HBM before the timed region (BASELINE.json configs[2]).  Data-parallel over ranks
(weak scaling: 64 images per GPU), one RCCL all-reduce of each flat gradient buffer per step.

Prints ONE JSON line (rank 0) with the driver contract fields plus
  "roofline":     in-situ HIP-event timing of the MFMA conv launches of one instrumented step: achieved
                  algorithmic TFLOP/s of the dominant kernel family vs the dense bf16 MFMA peak;
  "cpu_baseline": the CPU oracle (oracle/s2p_oracle.py, torch fp32 on the host cores) timed on a bounded sample of
                  the same workload -- a reported baseline, never the thing shipped.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16 peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--size", type=int, default=84)
    ap.add_argument("--env_type", type=str, default="cheetah")
    ap.add_argument("--precision", type=str, default="bf16")
    ap.add_argument("--no-graph", action="store_true", help="do not capture the step in a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=40)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--backend", type=str, default="", help="torch.distributed backend (default nccl = RCCL; gloo for a 1-GPU rehearsal)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: all ranks use GPU 0")
    ap.add_argument("--serial-streams", action="store_true",
                    help="run every launch on one stream (no overlap of independent chains): the configuration in which a "
                         "kernel's own duration is measured (rocprofv3 evidence for the roofline numbers); slower end to end")
    return ap.parse_args()


def spade_resblk_aggregate(recs, N, HW):
    """north_star's named target: all launches of the 6 SPADE/MAT ResBlks of ONE generator forward + backward
    (block convs, the fused gamma/beta and shared convs of the 12 MAT norms, the IN/MAT kernels).  The step runs
    the generator forward twice (G step, D step), so forward-kind launches are halved."""
    h = HW // 4
    def ms(r): return r["ms"]
    conv_shapes = {(N, h, h, 256, 256, 3, 1, 1, 0), (N, h, h, 256, 256, 3, 1, 12, 0), (N, h, h, 128, 512, 3, 1, 12, 0),
                   (N, h, h, 3, 1536, 3, 1, 1, 0)}      # groups == 12: the batched weight-gradient launches
    flops = t_mfma = t_norm = nbytes = 0.0
    launches = 0
    for r in recs:
        k = r["kind"]
        if k in ("fwd", "dgrad", "wgrad") and tuple(r["shape"]) in conv_shapes and r.get("net") == "G":     # (VGG conv3_x has the ResBlk conv's shape)
            w = 0.5 if k == "fwd" else 1.0
            flops += w * r["flops"]; t_mfma += w * ms(r); launches += w
        elif k.startswith("norm") and tuple(r["shape"][:4]) == (N, h, h, 256) and r["shape"][4] == 1 or \
                (k == "norm_stats" and tuple(r["shape"][:4]) == (N, h, h, 256)):
            w = 1.0 if k == "norm_bwd" else 0.5
            nbytes += w * r["bytes"]; t_norm += w * ms(r); launches += w
    if t_mfma <= 0:
        return None
    tot = t_mfma + t_norm
    return dict(gflop=round(flops / 1e9, 1), ms=round(tot, 3), mfma_ms=round(t_mfma, 3), norm_ms=round(t_norm, 3),
                launches=int(launches), tflops=round(flops / (tot * 1e-3) / 1e12, 2),
                frac=round(flops / (tot * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                mfma_only_frac=round(flops / (t_mfma * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                norm_hbm_tbps=round(nbytes / (t_norm * 1e-3) / 1e12, 2) if t_norm > 0 else None,
                includes="generator launches only (records are tagged with their network; VGG conv3_x, same shape as the block convs, is excluded). "
                         "6 blocks: 12 conv3x3 256->256 (fwd, dgrad, wgrad), grouped gamma/beta conv 12x(128->512) (fwd, dgrad, wgrad), "
                         "shared conv 3->1536 (fwd, wgrad), the 12 MAT norms forward and backward -- 11 forward and all 12 backward norms run "
                         "inside conv launches (s2p_conv2d_fwd_mat / s2p_conv2d_dgrad_mat: their HBM-bound tails are in mfma_ms), norm_ms is "
                         "what is left as separate launches; the fp32 state affine is excluded")


def hbm_rows(batch, size, version):
    """HBM side of the roofline, from the evidence committed under profiles/ (counters cannot be collected inside this process):
    per (kernel, grid) the HBM bytes per launch of the latest rocprofv3 --pmc passes (FETCH_SIZE x 2 per MI355X_MICROARCH.md, WRITE_SIZE)
    divided by that kernel's serial-stream dispatch time from the kernel trace of the same round.  `rows`: the three kernels without
    matrix work (MFMA-busy share < 2 %) that cost the step the most; `read_amplification`: MFMA kernels known to re-read their input
    (HBM read bytes against the input tensor's size).  Reported only while the files were measured on the loaded library version."""
    import csv, glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_traffic.json")))
    if not cands:
        return None
    tj = json.load(open(cands[-1]))
    meta = tj.get("_meta", {})
    if meta.get("s2p_version") != version:
        raise RuntimeError("%s was measured on library version %s, this run loads %s: hbm rows not reported" % (
            os.path.basename(cands[-1]), meta.get("s2p_version"), version))
    summ = cands[-1].replace("_pmc_traffic.json", "_serial_kernel_summary.csv")
    pmc = {}
    for k, v in tj.items():
        if k != "_meta" and "hbm_read_bytes" in v and "hbm_write_bytes" in v:
            n, g_ = [x.strip() for x in k.split("|")][:2]
            pmc[(n, g_.replace("grid ", ""))] = v
    joined = []
    step_bytes = step_us = 0.0
    for r in csv.DictReader(open(summ)):
        gx = [int(x) for x in r["grid"].split("x")]
        v = pmc.get((r["kernel"], str(gx[0]))) or pmc.get((r["kernel"], str(gx[0] * gx[1] * gx[2])))
        if v is None:
            continue
        by, us = v["hbm_read_bytes"] + v["hbm_write_bytes"], float(r["avg_us"])
        step_bytes += by * float(r["calls_per_step"]); step_us += float(r["us_per_step"])
        joined.append(dict(kernel=r["kernel"][:80], grid=r["grid"], us_per_step=float(r["us_per_step"]), avg_us=us,
                           hbm_read_mb=round(v["hbm_read_bytes"] / 1e6, 2), hbm_write_mb=round(v["hbm_write_bytes"] / 1e6, 2),
                           achieved_gbs=round(by / us / 1e3, 1), frac=round(by / us / 1e3 / HBM_PEAK_GBS, 4),
                           mfma_busy_share=round(v.get("mfma_util", 0.0), 4)))
    rows = sorted([j for j in joined if j["mfma_busy_share"] < 0.02 and j["hbm_read_mb"] + j["hbm_write_mb"] >= 8.0],
                  key=lambda j: -j["us_per_step"])[:3]
    # input re-reads of two MFMA kernels: VGG conv1_2 as 2-row bands, the decoder's transposed conv as four sub-pixel phases
    q = size // 2
    known = [("conv_planeg_kernel<3, 3, 3, 384, 6, 0, false, false>", str(batch * (size // 2) * 512) + "x1x1", batch * size * size * 64 * 2 / 1e6,
              "VGG conv1_2 64->64 at %dx%d as 2-row bands (input %dx%dx64)" % (size, size, size, size)),
             # (its grid: four phases x the pixel tiles of a phase rounded up to a multiple of 8, 256 threads each -- the 64-row weight
             # tile also serves the PatchGAN stride-2 dgrads, which must not be mistaken for it)
             ("conv_dma_kernel<64, 128, 2, 2>", "%dx1x1" % (4 * 8 * (((batch * q * q + 127) // 128 + 7) // 8) * 256), batch * q * q * 128 * 2 / 1e6,
              "decoder transposed conv 128->64 %d->%d (input %dx%dx128), four sub-pixel phases" % (q, size, q, q))]
    amp = []
    for name, grid, in_mb, what in known:
        m = [j for j in joined if j["kernel"] == name and (grid is None or j["grid"] == grid)]
        if m:
            j = max(m, key=lambda j: j["hbm_read_mb"])
            amp.append(dict(kernel=name, grid=j["grid"], layer=what, hbm_read_mb=j["hbm_read_mb"], input_mb=round(in_mb, 2),
                            read_amplification=round(j["hbm_read_mb"] / in_mb, 2), avg_us=j["avg_us"]))
    # the whole step: HBM-side bytes of every profiled kernel x its launches per step -- what the overlapped streams share
    whole = dict(gb_per_step=round(step_bytes / 1e9, 2), ms_at_peak=round(step_bytes / (HBM_PEAK_GBS * 1e9) * 1e3, 3),
                 serial_kernel_ms_covered=round(step_us / 1e3, 3),
                 note="sum over the kernels found in both files of (FETCH_SIZE x 2 + WRITE_SIZE) per launch x launches per step; the "
                      "counters sit on the L2's fabric side, so Infinity-Cache hits are included")
    return dict(peak=HBM_PEAK_GBS, unit="GB/s", rows=rows, read_amplification=amp, whole_step=whole,
                source="%s + %s" % (os.path.relpath(cands[-1], ROOT), os.path.relpath(summ, ROOT)), commit=meta.get("commit"),
                s2p_version=meta.get("s2p_version"))


def cpu_baseline(args, state_dim):
    """Oracle G+D train step (losses, both backward passes, both Adam updates) on the host cores, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import s2p_oracle as O
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    ncpu = max(1, min(ncpu, 16))          # the GPU box gives each GPU a 16-core share; more threads only thrash
    torch.set_num_threads(ncpu)
    print("[bench] cpu_baseline: oracle train step on %d host threads ..." % ncpu, file=sys.stderr, flush=True)
    spec = O.Spec(state_dim=state_dim)
    pg = O.init_params(O.generator_param_shapes(spec), 1)
    pd = O.init_params(O.discriminator_param_shapes(spec), 2)
    pv = O.init_params(O.vgg_param_shapes(), 3, kaiming=True)
    g = torch.Generator().manual_seed(1234)
    B = args.cpu_batch
    prev = torch.rand(B, 3, args.size, args.size, generator=g) * 2 - 1
    real = torch.rand(B, 3, args.size, args.size, generator=g) * 2 - 1
    state = torch.randn(B, state_dim, generator=g)
    mg = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in pg.items()}
    md = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in pd.items()}

    def step(t):
        for v in pg.values():
            v.requires_grad_(True); v.grad = None
        L, _ = O.generator_losses(pg, pd, pv, prev, state, real, spec)
        sum(L.values()).backward()
        with torch.no_grad():
            for k in pg:
                p, m, v = O.adam_step(pg[k].detach(), pg[k].grad, mg[k][0], mg[k][1], t, 1e-4, 0.0, 0.9)
                pg[k] = p; mg[k] = (m, v)
        for v in pd.values():
            v.requires_grad_(True); v.grad = None
        D = O.discriminator_losses(pg, pd, prev, state, real, spec)
        sum(D.values()).backward()
        with torch.no_grad():
            for k in pd:
                p, m, v = O.adam_step(pd[k].detach(), pd[k].grad, md[k][0], md[k][1], t, 4e-4, 0.0, 0.9)
                pd[k] = p; md[k] = (m, v)

    step(1)                                   # warm-up
    print("[bench] cpu_baseline: warm-up step done", file=sys.stderr, flush=True)
    t0 = time.time()
    for i in range(args.cpu_steps):
        step(2 + i)
        print("[bench] cpu_baseline: timed step %d done" % (i + 1), file=sys.stderr, flush=True)
    dt = time.time() - t0
    return dict(value=round(B * args.cpu_steps / dt, 3), unit="images/sec", cores=torch.get_num_threads(), kind="port",
                sample="oracle/s2p_oracle.py G+D train step, batch %d of the same %dx%d workload, %d timed steps "
                       "after 1 warm-up, torch fp32 on host cores" % (B, args.size, args.size, args.cpu_steps))


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves.  The parent has not touched the GPU (device_count
    does not initialise it) and never execs: torch.distributed.run is a fresh child process whose one JSON line and exit code
    are relayed."""
    import socket
    import subprocess
    if not args.share_gpu and torch.cuda.device_count() < args.gpus:
        print("bench.py --gpus %d: only %d HIP device(s) visible" % (args.gpus, torch.cuda.device_count()), file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC only on this pool (RCCL / cross-process sharing)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    elif r.returncode == 0:
        print("bench.py: the ranks printed no result line", file=sys.stderr)
        return 3
    return r.returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    # stdout carries exactly ONE JSON line (driver contract).  Native libraries write there too (RCCL prints a version banner
    # on communicator creation): point fd 1 at stderr for the run and keep the real stdout for the result line.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise RuntimeError("--gpus %d but the launcher started WORLD_SIZE %d ranks" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a HIP device: the S2P hot path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
        os.environ["LOCAL_RANK"] = "0"
    torch.cuda.set_device(local_rank)
    if world > 1 and args.backend:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.backend)

    from s2p_amd import ops
    ops.SERIALIZE = bool(args.serial_streams)
    from s2p_amd.options.train_options import TrainOptions
    from s2p_amd.trainers.pix2pix_trainer import Pix2PixTrainer

    opt = TrainOptions().parse(["--env_type", args.env_type, "--batchSize", str(args.batch), "--precision", args.precision,
                                "--gpu_ids", str(local_rank), "--checkpoints_dir", "/tmp/s2p_bench_ckpt",
                                "--crop_size", str(args.size)], quiet=True)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        trainer = Pix2PixTrainer(opt)
    dp = trainer.dp
    dev = torch.device("cuda", local_rank)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    data = dict(prev_image=(torch.rand(args.batch, 3, args.size, args.size, generator=g) * 2 - 1).to(dev),
                image=(torch.rand(args.batch, 3, args.size, args.size, generator=g) * 2 - 1).to(dev),
                state=torch.randn(args.batch, opt.state_dim, generator=g).to(dev))

    from s2p_amd.stepgraph import StepGraph
    model = trainer.pix2pix_model

    def train_step():
        trainer.run_generator_one_step(data)
        trainer.run_discriminator_one_step(data)

    # One step = both trainer calls.  It is captured into hipGraph segments by StepGraph: with one rank the whole step is a
    # single graph; with several ranks the trainer cuts the capture at every collective / event (the tail of the G gradient
    # is all-reduced under the rest of the G backward, D's all-reduce + Adam run under the next step's generator forward).
    use_graph = not args.no_graph
    sg = StepGraph(enabled=use_graph)
    trainer.seg = sg
    for _ in range(max(1, min(args.warmup, 2))):
        train_step()
    trainer.sync(); torch.cuda.synchronize()
    if use_graph:
        try:
            sg.capture(train_step)
            trainer.sync(); torch.cuda.synchronize()
        except Exception as e:     # capture not possible: fall back to eager launches and say so
            if rank == 0:
                print("hipGraph capture failed (%s); running eager" % str(e).splitlines()[0], file=sys.stderr)
            sg = StepGraph(enabled=False)
            trainer.seg = sg
            use_graph = False
            torch.cuda.synchronize()

    def step():
        if sg.captured:
            sg.replay()
        else:
            train_step()

    def eager_step():
        train_step()

    for _ in range(args.warmup):
        step()
    dp.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    trainer.sync(); torch.cuda.synchronize(); dp.barrier(); torch.cuda.synchronize()
    elapsed = dp.max_over_ranks(time.perf_counter() - t0)
    losses = {k: float(v.detach()) for k, v in trainer.get_latest_losses().items()}
    print("[bench] timed region done: %.3f ms/step" % (elapsed / args.steps * 1e3), file=sys.stderr, flush=True)
    if not all(v == v for v in losses.values()):
        raise RuntimeError("NaN in losses: %s" % losses)

    # ---- roofline leg: one instrumented eager step, events around every MFMA conv launch --------------------
    roofline = None
    comm_ms = []
    if not args.no_roofline:
        # every rank runs the instrumented step (it contains the gradient all-reduces); rank 0 reports
        # three instrumented steps; a launch's time is the MEDIAN of its three samples (one sample can be lengthened by a
        # hiccup -- a single step once showed the dominant launch at 51 us against 39-42 us in every other measurement)
        ops.SERIALIZE = True        # one serial chain of launches: a launch's event pair times that kernel alone
        reps = []
        for _ in range(3):
            ops.PROFILE = []
            trainer.comm_events = [] if dp.active else None
            eager_step()
            trainer.sync(); torch.cuda.synchronize()
            reps.append(ops.PROFILE)
            if dp.active:
                comm_ms.append(sum(e0.elapsed_time(e1) for e0, e1 in trainer.comm_events))
        trainer.comm_events = None
        ops.SERIALIZE = bool(args.serial_streams)
        ops.PROFILE = None
        all_recs = reps[0]
        same = all(len(r) == len(all_recs) for r in reps)
        for i, r in enumerate(all_recs):
            samples = [rep[i]["events"][0].elapsed_time(rep[i]["events"][1]) for rep in (reps if same else reps[:1])]
            r["ms"] = sorted(samples)[len(samples) // 2]
    if not args.no_roofline and rank == 0:
        recs = [r for r in all_recs if r["kind"] in ("fwd", "dgrad", "wgrad")]
        print("[bench] roofline leg done (%d conv launches, %d norm launches)" % (len(recs), len(all_recs) - len(recs)),
              file=sys.stderr, flush=True)
        resblk = spade_resblk_aggregate(all_recs, args.batch, 84)
        tot_f = sum(r["flops"] for r in recs)
        tot_ms = sum(r["ms"] for r in recs)
        by_kind = {}
        for r in recs:
            k = by_kind.setdefault(r["kind"], [0.0, 0.0, 0])
            k[0] += r["flops"]; k[1] += r["ms"]; k[2] += 1
        # dominant kernel: the implicit-GEMM conv family (conv_gather_kernel + wgrad_kernel), all launches of one step
        ach = tot_f / (tot_ms * 1e-3) / 1e12
        groups = {}
        for r in recs:
            gk = groups.setdefault((r["kind"], r["shape"]), [0.0, 0.0, 0])
            gk[0] += r["flops"]; gk[1] += r["ms"]; gk[2] += 1
        if os.environ.get("S2P_BENCH_LAYERS"):           # per-layer table of the instrumented step (stderr)
            for (gk, gshape), gv in sorted(groups.items(), key=lambda kv: -kv[1][1]):
                print("[bench] %-6s %-44s n=%2d  %7.3f ms  %6.0f TFLOP/s" % (gk, str(gshape), gv[2], gv[1], gv[0] / (gv[1] * 1e-3) / 1e12),
                      file=sys.stderr)
        (dk, dshape), dv = max(groups.items(), key=lambda kv: kv[1][1])
        dominant = dict(kind=dk, shape_N_H_W_Cin_Cout_k_stride_groups_transposed=list(dshape), launches_per_step=dv[2],
                        avg_launch_us=round(dv[1] * 1e3 / dv[2], 2), gflop_per_launch=round(dv[0] / dv[2] / 1e9, 2),
                        tflops=round(dv[0] / (dv[1] * 1e-3) / 1e12, 2), frac=round(dv[0] / (dv[1] * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4))
        # dominant KERNEL: the launch group (kind, shape, variant) with the largest time per step -- a shape runs as several
        # kernels (plain conv / + fused MAT-norm forward / + fused norm backward), each with its own duration and HBM bytes
        vgroups = {}
        for r in recs:
            vk = vgroups.setdefault((r["kind"], r["shape"], r.get("variant", "plain")), [0.0, 0.0, 0, 0.0])
            vk[0] += r["flops"]; vk[1] += r["ms"]; vk[2] += 1; vk[3] += r.get("norm_bytes", 0.0)
        (vkind, vshape, vvar), vv = max(vgroups.items(), key=lambda kv: kv[1][1])
        dominant_kernel = dict(kind=vkind, variant=vvar, shape_N_H_W_Cin_Cout_k_stride_groups_transposed=list(vshape),
                               launches_per_step=vv[2], us_per_step=round(vv[1] * 1e3, 1), avg_launch_us=round(vv[1] * 1e3 / vv[2], 2),
                               gflop_per_launch=round(vv[0] / vv[2] / 1e9, 2), tflops=round(vv[0] / (vv[1] * 1e-3) / 1e12, 2),
                               frac=round(vv[0] / (vv[1] * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                               fused_norm_mb_per_launch=round(vv[3] / vv[2] / 1e6, 2))
        # the dominant launch alone and SUSTAINED: 20 launches of the ResBlk conv captured in one hipGraph, replayed 5 times between
        # two HIP events (no host launch gap; back-to-back MFMA load, so the chip runs at its sustained MFMA clock: slower than
        # the same launch between the HBM-bound norm kernels of the real step, which is what rocprofv3's per-dispatch time shows)
        try:
            if world > 1:
                raise RuntimeError("skipped with more than one rank (no extra graph capture beside a live RCCL group)")
            lay = model.netG.lay["b0c0"]
            hq = args.size // 4
            xa = torch.randn(args.batch, hq, hq, 256, device=dev).to(model.netG.compute_dtype)
            for _ in range(3):
                lay.fwd(xa)
            torch.cuda.synchronize()
            gk = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gk, capture_error_mode="relaxed"):
                for _ in range(20):
                    lay.fwd(xa)
            gk.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                gk.replay()
            e1.record(); torch.cuda.synchronize()
            k_us = e0.elapsed_time(e1) * 1e3 / 100
            k_fl = 2.0 * args.batch * hq * hq * 256 * 256 * 9
            dominant["kernel_alone_us"] = round(k_us, 2)
            dominant["kernel_alone_tflops"] = round(k_fl / (k_us * 1e-6) / 1e12, 1)
            dominant["kernel_alone_frac"] = round(k_fl / (k_us * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)
        except Exception as e:      # the extra figure must never fail the bench
            print("[bench] kernel-alone leg: %s" % str(e).splitlines()[0], file=sys.stderr)
        # HBM traffic of the dominant launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE in separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md): counters cannot be collected from
        # inside this process, so the latest committed measurement of the same kernel + grid is reported, with its source
        traffic, traffic_detail = None, None
        try:
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_traffic.json")))
            if cands:
                tj = json.load(open(cands[-1]))
                want_grid = str(args.batch * 4 * 512)                 # plane-resident conv: one 512-thread workgroup per (image, 64-channel slab)
                meta = tj.get("_meta", {})
                from s2p_amd import _lib as _l
                if meta.get("s2p_version") != _l.lib().s2p_version():
                    raise RuntimeError("%s was measured on library version %s, this run loads %s: not reported" % (
                        os.path.basename(cands[-1]), meta.get("s2p_version"), _l.lib().s2p_version()))
                # the counters of the kernel variant that dominates the dominant shape in THIS run's instrumented step
                # (template argument MAT: 0 plain conv, 1 + fused norm forward, 2 + fused norm backward)
                hq = args.size // 4
                same_shape = {v: g_ for (k_, sh, v), g_ in vgroups.items() if sh == dshape and k_ in ("fwd", "dgrad")}
                tvar = max(same_shape, key=lambda v: same_shape[v][1]) if same_shape else "plain"
                mat = {"plain": 0, "mat_fwd": 1, "mat_bwd": 2}[tvar]
                tensor_mb = args.batch * hq * hq * 256 * 2 / 1e6
                alg_mb = {0: 2 * tensor_mb, 1: 5 * tensor_mb, 2: 8 * tensor_mb}[mat] + 256 * 2304 * 2 / 1e6
                for k, v in tj.items():
                    # (template arguments <PB, WP, DIAG, MAT[, GST]>: the staged form of round 5 prints a fifth one)
                    if (k.startswith("conv_plane_kernel<7, 22, 0, %d>" % mat) or k.startswith("conv_plane_kernel<7, 22, 0, %d," % mat)) \
                            and ("grid %s " % want_grid) in k and "hbm_read_bytes" in v:
                        traffic = int(v["hbm_read_bytes"] + v["hbm_write_bytes"])          # HBM bytes per launch of the dominant kernel
                        traffic_detail = dict(variant=tvar, hbm_read_mb=round(v["hbm_read_bytes"] / 1e6, 2), hbm_write_mb=round(v["hbm_write_bytes"] / 1e6, 2),
                                       algorithmic_mb=round(alg_mb, 2),
                                       algorithmic_is="x + y (+ gamma, beta, modulated y | + xn, gamma, beta, res, dxn, dgamma, dbeta) + weights",
                                       mfma_busy_share=round(v.get("mfma_util", 0.0), 4), launches_profiled=v.get("launches"),
                                       source=os.path.relpath(cands[-1], ROOT) + " :: " + k, commit=meta.get("commit"),
                                       s2p_version=meta.get("s2p_version"))
                        break
        except Exception as e:      # evidence file unreadable: report null rather than fail the bench
            print("[bench] traffic: %s" % e, file=sys.stderr)
        roofline = dict(bound="mfma", measured="three instrumented eager steps with every launch on ONE stream (HIP events around each "
                                               "launch, median of the three samples per launch): kernel-alone durations; rocprofv3 counterpart: bench.py --serial-streams",
                        kernel="conv family: conv_plane / conv_halo / conv_dma / conv_gather (fwd, dgrad, incl. fused norm tails) + wgrad_slab / "
                                             "wgrad_dma / wgrad + thin kernels, all %d launches of one step" % len(recs),
                        dominant_layer=dominant, dominant_kernel=dominant_kernel, spade_resblk_fwd_bwd=resblk,
                        achieved=round(ach, 2), peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s", frac=round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                        frac_is="algorithmic_gflop_per_step / conv_ms_per_step (serial-stream sum of the conv family's kernel time) / peak",
                        traffic=traffic, traffic_detail=traffic_detail, algorithmic_gflop_per_step=round(tot_f / 1e9, 1), conv_ms_per_step=round(tot_ms, 3),
                        avg_launch_us=round(tot_ms * 1e3 / max(len(recs), 1), 2),
                        by_kind={k: dict(gflop=round(v[0] / 1e9, 1), ms=round(v[1], 3), launches=v[2],
                                         tflops=round(v[0] / (v[1] * 1e-3) / 1e12, 2)) for k, v in by_kind.items()})
        try:
            from s2p_amd import _lib as _l2
            roofline["hbm_rows"] = hbm_rows(args.batch, args.size, _l2.lib().s2p_version())
        except Exception as e:      # evidence files absent / of another library version: null, never a failed bench
            print("[bench] hbm_rows: %s" % e, file=sys.stderr)
            roofline["hbm_rows"] = None

    devices = sorted(set(dp.all_gather_ints(torch.cuda.current_device()))) if dp.active else None      # (a collective: every rank)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, opt.state_dim)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        if roofline is not None:      # the whole overlapped step against the MFMA peak (neither the dominant kernel nor the serial family sum)
            roofline["step_frac"] = round(roofline["algorithmic_gflop_per_step"] / ms / MFMA_BF16_PEAK_TFLOPS, 4)
        out = {
            "metric": "G+D train-step images/sec at 84x84 bs=64/GPU",
            "value": round(args.batch * world * args.steps / elapsed, 2),
            "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "configs[2]: full G+D train step (hinge GAN + feature matching + VGG19 perceptual "
                                   "+ L1, Adam x2), %dx%d %s, netG=s2p, netD=multiscale(2)" % (args.size, args.size, args.env_type),
                       "global_batch": args.batch * world, "per_gpu_batch": args.batch,
                       "parallelism": "dp%d" % world, "hip_graph": bool(use_graph), "graph_segments": sg.n_graphs,
                       "streams": "serial" if args.serial_streams else "overlapped",
                       "vgg_weights": "seeded stand-in (ImageNet weights unobtainable offline)"},
            "roofline": roofline, "cpu_baseline": cpu, "losses": {k: round(v, 4) for k, v in losses.items()},
        }
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            # what the ranks actually ran on: backend ("nccl" = RCCL on ROCm), the size of the group the collectives used, and the
            # time the compute stream spent WAITING for the communication stream in an instrumented eager step (events around
            # the two join points: end of the G exchange, first use of D in the next G step)
            out["distributed"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                  "devices": devices,
                                  "exposed_comm_ms_per_step": round(sorted(comm_ms)[len(comm_ms) // 2], 3) if comm_ms else None}
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():       # orderly RCCL teardown: every rank leaves together
        trainer.sync(); torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
