mkdir -p gpurun_out/r05
for abl in 0 1 2 4 8 3 6 7 15; do
  v=$((1 + 16*abl))
  echo "== ABL $abl (AGB_CMPT=$v)"
  AGB_CMPT=$v python tools/bench_conv.py --modes 128 --reps 20 2>&1 | grep "^ts"
done
echo "== old kernel"
AGB_CMPT=0 python tools/bench_conv.py --modes 128 --reps 20 2>&1 | grep "^ts"
