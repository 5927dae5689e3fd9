#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench10.log 2>&1
grep "timed region\|^{" gpurun_out/bench10.log | cut -c1-330
# python-level profile of the host side of 20 steps
timeout 300 python - > gpurun_out/cprofile10.log 2>&1 <<'PY'
import cProfile, pstats, sys, io, os
sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"]
sys.path.insert(0, os.getcwd())
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
PY
tail -75 gpurun_out/cprofile10.log | cut -c1-200
