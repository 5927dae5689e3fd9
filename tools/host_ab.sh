mkdir -p gpurun_out/r05
for mode in hold record hold record; do
  if [ $mode = record ]; then export AGB_INPUT_RECORD_STREAM=1; else unset AGB_INPUT_RECORD_STREAM; fi
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs > gpurun_out/r05/host_$mode.json 2> gpurun_out/r05/host_$mode.log
  echo "== $mode"; grep -E "CPU time per step|host CPU" gpurun_out/r05/host_$mode.log | cut -c1-400
  python -c "import json;d=json.load(open('gpurun_out/r05/host_$mode.json'));print(d['value'],d['ms_per_step'],d['host_enqueue_ms_p50'],d['host_cpu_ms_per_step_p50'],d['host_main_thread_cpu_ms_p50'])"
done
