#!/bin/bash
# bench.py with and without --maintenance at 65 536 and 32 768 plants -> gpurun_out/r3/bench_*.json, one summary line each
mkdir -p gpurun_out/r3
for n in 65536 32768; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --maintenance --plants-per-gpu $n > gpurun_out/r3/bench_maint_$n.json 2> gpurun_out/r3/bench_maint_$n.err
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --plants-per-gpu $n > gpurun_out/r3/bench_$n.json 2> gpurun_out/r3/bench_$n.err
done
python - <<PY
import json
for f in ("bench_maint_65536","bench_65536","bench_maint_32768","bench_32768"):
    try:
        d=json.load(open("gpurun_out/r3/%s.json"%f)); print(f, "ms/step %.4f kernel_ms %.4f selfcheck %.4f"%(d["ms_per_step"], d["roofline"]["kernel_ms"], d["selfcheck"]["ms_per_step"]), d["roofline"]["kernel"])
    except Exception as e: print(f,"ERR",e)
PY
