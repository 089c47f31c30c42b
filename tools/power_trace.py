"""Socket power and shader clock while the captured G+D train step replays back to back (is the step power-limited?).

A sampler thread reads the amdgpu hwmon files of every card the box shows (power1_average / power1_input in microwatts,
freq1_input in Hz; an ordinary user may read them) every 20 ms during: 2 s idle, N replays of the step, 2 s idle.  The card
whose power moves is ours.  Where the hwmon files are absent the script says so and polls `rocm-smi --json` instead (5 Hz).
    python tools/power_trace.py [steps]                 -> gpurun_out/power_trace.json + a summary on stdout"""
import os, sys, time, glob, json, threading, subprocess, io, contextlib
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402
import torch
from s2p_amd.options.train_options import TrainOptions
from s2p_amd.trainers.pix2pix_trainer import Pix2PixTrainer
from s2p_amd.stepgraph import StepGraph

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 800


def hwmon_files():
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
        for hw in glob.glob(os.path.join(card, "device/hwmon/hwmon*")):
            f = {}
            for name in ("power1_average", "power1_input", "freq1_input", "freq2_input", "power1_cap", "temp1_input"):
                p = os.path.join(hw, name)
                if os.path.exists(p):
                    f[name] = p
            if f:
                out[os.path.basename(card)] = f
    return out


def read_int(p):
    try:
        with open(p) as fh:
            return int(fh.read().strip())
    except Exception:
        return None


class Sampler(threading.Thread):
    def __init__(self, files, dt=0.02):
        super().__init__(daemon=True)
        self.files, self.dt, self.rows, self.stop = files, dt, [], False

    def run(self):
        while not self.stop:
            t = time.perf_counter()
            row = {"t": t}
            for card, f in self.files.items():
                pw = read_int(f.get("power1_average") or f.get("power1_input") or "")
                row[card] = (pw, read_int(f.get("freq1_input", "")))
            self.rows.append(row)
            time.sleep(self.dt)


class SmiSampler(threading.Thread):
    def __init__(self, dt=0.2):
        super().__init__(daemon=True)
        self.dt, self.rows, self.stop = dt, [], False

    def run(self):
        while not self.stop:
            t = time.perf_counter()
            try:
                js = json.loads(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True,
                                               text=True, timeout=5).stdout)
            except Exception as e:
                js = {"error": str(e)}
            self.rows.append({"t": t, "smi": js})
            time.sleep(self.dt)


files = hwmon_files()
print("hwmon:", {c: sorted(f) for c, f in files.items()}, flush=True)
for c, f in files.items():
    if "power1_cap" in f:
        print(c, "power cap", read_int(f["power1_cap"]), "uW", flush=True)

opt = TrainOptions().parse(["--env_type", "cheetah", "--batchSize", "64", "--precision", "bf16", "--gpu_ids", "0",
                            "--checkpoints_dir", "/tmp/pw_ck"], quiet=True)
with contextlib.redirect_stdout(io.StringIO()):
    tr = Pix2PixTrainer(opt)
g = torch.Generator().manual_seed(0)
data = dict(prev_image=(torch.rand(64, 3, 84, 84, generator=g) * 2 - 1).cuda(), image=(torch.rand(64, 3, 84, 84, generator=g) * 2 - 1).cuda(),
            state=torch.randn(64, 17, generator=g).cuda())


def step():
    tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)


for _ in range(2):
    step()
torch.cuda.synchronize()
sg = StepGraph(); tr.seg = sg
sg.capture(step)
for _ in range(5):
    sg.replay()
torch.cuda.synchronize()

smp = Sampler(files) if files else SmiSampler()
smp.start()
time.sleep(2.0)
marks = []
# the run in chunks of 100 replays: ms/step of every chunk shows whether the step slows as the package heats up
t_begin = time.perf_counter()
for c in range(max(1, steps // 100)):
    t0 = time.perf_counter()
    for _ in range(100):
        sg.replay()
    torch.cuda.synchronize()
    marks.append((t0, time.perf_counter()))
t_end = time.perf_counter()
time.sleep(2.0)
smp.stop = True
smp.join()

ms = [(b - a) / 100 * 1e3 for a, b in marks]
print("ms/step per 100 replays: first %.3f  median %.3f  last %.3f" % (ms[0], sorted(ms)[len(ms) // 2], ms[-1]))
summary = {"steps": len(marks) * 100, "ms_per_step_chunks": [round(x, 4) for x in ms]}
if files:
    for card in files:
        idle = [r[card] for r in smp.rows if r["t"] < t_begin - 0.2 and r[card][0] is not None]
        busy = [r[card] for r in smp.rows if t_begin + 1.0 < r["t"] < t_end and r[card][0] is not None]
        if not idle or not busy:
            continue
        pw_i = sum(p for p, _ in idle) / len(idle) / 1e6
        pw_b = sum(p for p, _ in busy) / len(busy) / 1e6
        pw_max = max(p for p, _ in busy) / 1e6
        fr = [f for _, f in busy if f]
        fr_i = [f for _, f in idle if f]
        cap = read_int(files[card].get("power1_cap", ""))
        summary[card] = {"idle_W": round(pw_i, 1), "busy_W": round(pw_b, 1), "busy_max_W": round(pw_max, 1),
                         "cap_W": cap / 1e6 if cap else None,
                         "busy_sclk_MHz": [round(min(fr) / 1e6), round(sum(fr) / len(fr) / 1e6), round(max(fr) / 1e6)] if fr else None,
                         "idle_sclk_MHz": round(sum(fr_i) / len(fr_i) / 1e6) if fr_i else None, "samples": len(busy)}
        print(card, summary[card])
else:
    summary["smi_first"] = smp.rows[0] if smp.rows else None
    summary["smi_mid"] = smp.rows[len(smp.rows) // 2] if smp.rows else None
    print(json.dumps(summary["smi_mid"])[:2000])
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
with open(os.path.join(R, "gpurun_out", "power_trace.json"), "w") as fh:
    json.dump(summary, fh, indent=1)
