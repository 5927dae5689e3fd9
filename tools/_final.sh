cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/final13
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final13/smoke.log 2>&1; tail -1 gpurun_out/final13/smoke.log
python bench.py > gpurun_out/final13/bench.json 2> gpurun_out/final13/bench.err; head -c 330 gpurun_out/final13/bench.json; echo
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final13/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/final13/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/final13/prof_bench.err)
head -c 330 gpurun_out/final13/prof_bench.json; echo
bash tools/collect_pmc.sh v13 > gpurun_out/final13/pmc.log 2>&1; tail -2 gpurun_out/final13/pmc.log
