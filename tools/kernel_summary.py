"""Per-dispatch summary of a rocprofv3 --kernel-trace run: one row per (kernel, grid size, workgroup size, LDS size), so
launches of one kernel template on different problem shapes are NOT averaged together (rocprofv3 --stats averages per
kernel name).  Usage: python tools/kernel_summary.py <dir with *_kernel_trace.csv> <out.csv> [steps]
With `steps` (number of timed + warm-up bench steps replayed under the profiler) the per-step launch count and time are
also given.  No GPU needed."""
import collections, csv, glob, os, re, sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    return name[:110]


def main():
    src, out = sys.argv[1], sys.argv[2]
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else None
    files = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit("no *kernel_trace.csv under " + src)
    agg = collections.OrderedDict()
    for f in files:
        for r in csv.DictReader(open(f)):
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            grid = "x".join(str(r.get(k, "1")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")) if "Grid_Size_X" in r else r.get("Grid_Size", "")
            wg = "x".join(str(r.get(k, "1")) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z")) if "Workgroup_Size_X" in r else r.get("Workgroup_Size", "")
            key = (short(r["Kernel_Name"]), grid, wg, r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""))
            a = agg.setdefault(key, [0, 0.0, 1e30, 0.0])
            a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    total = sum(v[1] for _, v in rows)
    with open(out, "w", newline="") as fo:
        w = csv.writer(fo)
        hdr = ["kernel", "grid", "workgroup", "lds_bytes", "vgprs", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"]
        if steps:
            hdr += ["calls_per_step", "us_per_step"]
        w.writerow(hdr)
        for (k, grid, wg, lds, vg), (n, t, mn, mx) in rows:
            row = [k, grid, wg, lds, vg, n, round(t, 1), round(t / n, 2), round(mn, 2), round(mx, 2), round(100 * t / total, 2)]
            if steps:
                row += [round(n / steps, 2), round(t / steps, 1)]
            w.writerow(row)
    print("%d dispatch groups, %.1f us total kernel time -> %s" % (len(rows), total, out))


if __name__ == "__main__":
    main()
