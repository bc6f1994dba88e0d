#!/bin/bash
# Round-3 (second session) evidence on the final code: bench lines + rocprofv3 kernel statistics of the same commands.
# usage (GPU box, repo root): bash profiles/collect_r03b.sh     (outputs: gpurun_out/r03b/, copied to profiles/r03b_* by hand)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b
mkdir -p $O
python3 $R/bench.py --roofline-rows 0 > $O/bench_all.json 2> $O/bench_all.err
echo bench_all done
python3 $R/bench.py --workload hires --steps 20 > $O/bench_hires.json 2> $O/bench_hires.err
python3 $R/bench.py --workload hires-train --steps 20 > $O/bench_hires_train.json 2> $O/bench_hires_train.err
python3 $R/bench.py --workload train --da --steps 20 --no-cpu-baseline --no-roofline-top > $O/bench_train_da.json 2> $O/bench_da.err
echo benches done
python3 $R/profiles/segment_timeline.py 2>&1 | grep -v amdgpu.ids > $O/segment_timeline.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_roof -o roof -- python3 $R/bench.py --roofline-only > $O/roofline_only.json 2> $O/prof_roof.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o train -- python3 $R/bench.py --workload train --no-cpu-baseline --no-roofline-top --no-parity --steps 50 > $O/prof_train.json 2> $O/prof_train.log
echo prof_train done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fwd -o fwd -- python3 $R/bench.py --workload fwd --no-cpu-baseline --no-parity --steps 50 > $O/prof_fwd.json 2> $O/prof_fwd.log
python3 $R/profiles/step_timeline.py $O/prof_train > $O/step_timeline.txt 2>&1
# keep what travels back small: statistics only, no traces
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
ls $O
