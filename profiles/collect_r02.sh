#!/bin/bash
# Round-2 evidence behind DESIGN.md section 5: bench lines + rocprofv3 kernel stats of the same commands.
# usage (GPU box, repo root): bash profiles/collect_r02.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02
mkdir -p $O
python3 $R/bench.py > $O/bench_all.json 2> $O/bench_all.err
python3 $R/bench.py --workload hires --steps 20 > $O/bench_hires.json 2> $O/bench_hires.err
python3 $R/bench.py --workload train --da --steps 20 --no-cpu-baseline > $O/bench_train_da.json 2> $O/bench_da.err
python3 $R/bench.py --workload train --da all --steps 20 --no-cpu-baseline > $O/bench_train_da_all.json 2> $O/bench_da_all.err
python3 $R/bench.py --workload fwd --da all --steps 50 --no-cpu-baseline > $O/bench_fwd_da_all.json 2> $O/bench_fwd_da_all.err
python3 $R/profiles/microbench_fc_update.py > $O/microbench_fc_update.txt 2>&1
python3 $R/profiles/microbench_da.py 2>&1 | grep -v amdgpu.ids > $O/microbench_da.txt
python3 $R/profiles/segment_timeline.py 2>&1 | grep -v amdgpu.ids > $O/segment_timeline.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_roof -o roof -- python3 $R/bench.py --roofline-only > $O/roofline_only.json 2> $O/prof_roof.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o train -- python3 $R/bench.py --workload train --no-cpu-baseline --steps 50 > $O/prof_train.json 2> $O/prof_train.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fwd -o fwd -- python3 $R/bench.py --workload fwd --no-cpu-baseline --steps 50 > $O/prof_fwd.json 2> $O/prof_fwd.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hires -o hires -- python3 $R/bench.py --workload hires --steps 10 > $O/prof_hires.json 2> $O/prof_hires.log
ls $O
