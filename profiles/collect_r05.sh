#!/bin/bash
# Round-5 evidence on the current code.  usage (GPU box, repo root): bash profiles/collect_r05.sh TAG STAGE...
#   stages: tests | bench | timelines | prof_train | prof_fwd | prof_hires | pmc_step | hires | pk
# outputs: gpurun_out/TAG/ (copied to profiles/r05_* by hand).  Every profile is taken from `bench.py --steps-only`, i.e. a
# process that runs bench-mode steps and nothing else (round 4's statistics mixed 15 fp32-class steps and the roofline
# microbenchmarks into the same CSV); step_timeline.py asserts that its window holds no fp32-class instantiation.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r05}; shift; O=$R/gpurun_out/$TAG
mkdir -p $O
for stage in "$@"; do
case $stage in
tests)
  (cd $R && python3 -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; tail -3 $O/pytest.txt) ;;
tests_s)   # the tests whose printed measurements the tolerances quote
  (cd $R && python3 -m pytest tests/test_fullsize_gpu.py tests/test_hires_gpu.py -x -q -s -m gpu -k "bench_mode or b2_against or gradients_match_oracle" > $O/pytest_s.txt 2>&1; tail -3 $O/pytest_s.txt) ;;
bench)
  python3 $R/bench.py --roofline-rows 0 > $O/bench_all.json 2> $O/bench_all.err && echo bench_all done ;;
bench_quick)
  python3 $R/bench.py --steps 50 --no-cpu-baseline --no-parity --no-roofline-top > $O/bench_quick.json 2> $O/bench_quick.err && python3 -c "import json;d=json.load(open('$O/bench_quick.json'));print('step',d['ms_per_step'],'fwd',d['fwd']['ms_per_step'])" ;;
timelines)
  python3 $R/profiles/segment_timeline.py 2>&1 | grep -v amdgpu.ids > $O/segment_timeline.txt
  python3 $R/profiles/fwd_branches.py 2>&1 | grep -v amdgpu.ids > $O/fwd_branches.txt; cat $O/fwd_branches.txt ;;
prof_train)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o train -- python3 $R/bench.py --workload train --steps-only --steps 50 > $O/prof_train.json 2> $O/prof_train.log
  python3 $R/profiles/step_timeline.py $O/prof_train > $O/step_timeline.txt 2>&1; head -3 $O/step_timeline.txt
  cp $(find $O/prof_train -name "*kernel_stats.csv" | head -1) $O/train_b32_kernel_stats.csv
  (cd $R && python3 - <<PY > $O/train_b32_kernel_stats.csv.json
import json, subprocess
print(json.dumps({"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload train --steps-only --steps 50", "batch": 32,
                  "distortion_aware": [], "commit": open("$R/.gpurun_commit").read().strip() if __import__("os").path.exists("$R/.gpurun_commit") else "working tree of the round-5 session"}))
PY
) ;;
prof_fwd)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fwd -o fwd -- python3 $R/bench.py --workload fwd --steps-only --steps 50 > $O/prof_fwd.json 2> $O/prof_fwd.log
  python3 $R/profiles/fwd_timeline.py $O/prof_fwd > $O/fwd_timeline.txt 2>&1; head -3 $O/fwd_timeline.txt
  cp $(find $O/prof_fwd -name "*kernel_stats.csv" | head -1) $O/fwd_b32_kernel_stats.csv ;;
pmc_step)
  bash $R/profiles/pmc_step_traffic.sh > $O/pmc_step_traffic.txt 2>&1; head -3 $O/pmc_step_traffic.txt ;;
hires)
  python3 $R/bench.py --workload hires --steps 20 > $O/bench_hires.json 2> $O/bench_hires.err
  python3 $R/bench.py --workload hires-train --steps 20 > $O/bench_hires_train.json 2> $O/bench_hires_train.err
  python3 $R/bench.py --workload hires-train --da res,decoders --steps 20 > $O/bench_hires_train_da.json 2> $O/bench_hires_train_da.err
  python3 $R/bench.py --workload train --da --steps 20 --no-cpu-baseline --no-roofline-top --no-parity > $O/bench_train_da.json 2> $O/bench_da.err ;;
prof_hires)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hires -o hires -- python3 $R/bench.py --workload hires-train --steps 30 --warmup 3 --no-cpu-baseline --no-parity --no-roofline-top > $O/prof_hires.json 2> $O/prof_hires.log
  cp $(find $O/prof_hires -name "*kernel_stats.csv" | head -1) $O/hires_train_b8_kernel_stats.csv; head -12 $O/hires_train_b8_kernel_stats.csv | cut -c1-200 ;;
pk)
  (cd $R/profiles/experiments/pk_hazard && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 pk_opsel.hip -o /tmp/pk_opsel && timeout -k 10 240 /tmp/pk_opsel > $O/pk_opsel_erratum.txt 2>&1; tail -3 $O/pk_opsel_erratum.txt) ;;
esac
done
# keep what travels back small: statistics only, no traces
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -delete
ls $O
