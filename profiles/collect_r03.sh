#!/bin/bash
export HDRSKY_EXPERIMENTS=1   # the tuning hooks this script sets are behind the gate since round 4 (csrc/hooks.h, hooks.py)
# Round-3 evidence behind DESIGN.md section 5: bench lines + rocprofv3 kernel stats of the same commands.
# usage (GPU box, repo root): bash profiles/collect_r03.sh     (outputs: gpurun_out/r03/, copied to profiles/r03_* by hand)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03
mkdir -p $O
python3 $R/bench.py --roofline-rows 0 > $O/bench_all.json 2> $O/bench_all.err
python3 $R/bench.py --workload hires --steps 20 > $O/bench_hires.json 2> $O/bench_hires.err
python3 $R/bench.py --workload hires-train --steps 20 > $O/bench_hires_train.json 2> $O/bench_hires_train.err
python3 $R/bench.py --workload hires-train --da res,decoders --steps 20 > $O/bench_hires_train_da.json 2> $O/bench_hires_train_da.err
python3 $R/bench.py --workload train --da --steps 20 --no-cpu-baseline --no-roofline-top > $O/bench_train_da.json 2> $O/bench_da.err
python3 $R/bench.py --workload train --da all --steps 20 --no-cpu-baseline --no-roofline-top > $O/bench_train_da_all.json 2> $O/bench_da_all.err
HDRSKY_BENCH_FORCE_DP=1 python3 $R/bench.py --workload train --steps 20 --no-cpu-baseline --no-roofline-top > $O/bench_dp1_rccl_world1.json 2> $O/bench_dp1.err
HDRSKY_BENCH_ONE_CARD=1 HDRSKY_DIST_BACKEND=gloo python3 $R/bench.py --gpus 2 --workload train --steps 5 --warmup 2 --no-cpu-baseline --no-roofline-top > $O/bench_dp2_one_card_gloo.json 2> $O/bench_dp2.err
python3 $R/profiles/segment_timeline.py 2>&1 | grep -v amdgpu.ids > $O/segment_timeline.txt
python3 $R/profiles/microbench_wgrad2.py 2>&1 | grep -v amdgpu.ids > $O/microbench_wgrad2.txt
python3 $R/profiles/microbench_wgrad2.py --group 2>&1 | grep -v amdgpu.ids >> $O/microbench_wgrad2.txt
python3 $R/profiles/stamp_wgrad2.py 2>&1 | grep -v amdgpu.ids > $O/stamp_wgrad2.txt
python3 $R/profiles/stamp_wgrad3.py 2>&1 | grep -v amdgpu.ids > $O/stamp_wgrad3.txt
python3 $R/profiles/microbench_dog.py 2>&1 | grep -v amdgpu.ids > $O/microbench_dog.txt
HDRSKY_DISC_SPLIT=1 python3 $R/profiles/segment_timeline.py 2>&1 | grep -v amdgpu.ids > $O/segment_timeline_disc_split.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_roof -o roof -- python3 $R/bench.py --roofline-only > $O/roofline_only.json 2> $O/prof_roof.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o train -- python3 $R/bench.py --workload train --no-cpu-baseline --no-roofline-top --no-parity --steps 50 > $O/prof_train.json 2> $O/prof_train.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fwd -o fwd -- python3 $R/bench.py --workload fwd --no-cpu-baseline --no-parity --steps 50 > $O/prof_fwd.json 2> $O/prof_fwd.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hires_train -o hires_train -- python3 $R/bench.py --workload hires-train --steps 10 > $O/prof_hires_train.json 2> $O/prof_hires_train.log
python3 $R/profiles/step_timeline.py $O/prof_train > $O/step_timeline.txt 2>&1
# keep what travels back small: statistics only, no traces
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
ls $O
