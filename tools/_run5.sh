set -e
python -m pytest tests/test_model_gpu.py -x -q -k "small_width" > gpurun_out/r5_t5.log 2>&1 || { tail -40 gpurun_out/r5_t5.log; exit 1; }
tail -2 gpurun_out/r5_t5.log
python bench.py --steps 30 --warmup 5 > gpurun_out/r5_bench_b.json 2> gpurun_out/r5_bench_b.err || { tail -20 gpurun_out/r5_bench_b.err; exit 1; }
python - <<'PY'
import json
j = json.load(open("gpurun_out/r5_bench_b.json"))
r = j["roofline"]
print(j["value"], j["ms_per_step"], "resblk", r["spade_resblk_fwd_bwd"]["frac"], r["spade_resblk_fwd_bwd"]["ms"], r["spade_resblk_fwd_bwd"]["gflop"], "dominant_kernel", r["dominant_kernel"]["variant"], r["dominant_kernel"]["avg_launch_us"], r["dominant_kernel"]["frac"], "step_frac", r["step_frac"], "hbm_rows", r["hbm_rows"], "traffic", r["traffic"])
PY
