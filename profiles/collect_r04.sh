#!/bin/bash
# Round-4 evidence on the current code: bench lines + rocprofv3 kernel statistics of the same commands + timelines + the HBM
# traffic of the roofline kernel (FETCH_SIZE / WRITE_SIZE in separate --pmc passes with --kernel-trace only).
# usage (GPU box, repo root): bash profiles/collect_r04.sh [tag]   (outputs: gpurun_out/<tag>/, copied to profiles/r04_* by hand)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04}
mkdir -p $O
python3 $R/bench.py --roofline-rows 0 > $O/bench_all.json 2> $O/bench_all.err && echo bench_all done
python3 $R/bench.py --workload hires --steps 20 > $O/bench_hires.json 2> $O/bench_hires.err
python3 $R/bench.py --workload hires-train --steps 20 > $O/bench_hires_train.json 2> $O/bench_hires_train.err
python3 $R/bench.py --workload hires-train --da res,decoders --steps 20 > $O/bench_hires_train_da.json 2> $O/bench_hires_train_da.err
python3 $R/bench.py --workload train --da --steps 20 --no-cpu-baseline --no-roofline-top > $O/bench_train_da.json 2> $O/bench_da.err
HDRSKY_BENCH_FORCE_DP=1 python3 $R/bench.py --workload train --steps 20 --no-cpu-baseline --no-roofline-top --no-parity > $O/bench_dp1_rccl_world1.json 2> $O/bench_dp1.err
HDRSKY_BENCH_ONE_CARD=1 HDRSKY_DIST_BACKEND=gloo python3 $R/bench.py --gpus 2 --workload train --steps 5 --warmup 2 --no-cpu-baseline --no-roofline-top --no-parity > $O/bench_dp2_one_card_gloo.json 2> $O/bench_dp2.err
python3 $R/bench.py --workload train --da all --steps 20 --no-cpu-baseline --no-roofline-top --no-parity > $O/bench_train_da_all.json 2> $O/bench_da_all.err
echo benches done
python3 $R/profiles/segment_timeline.py 2>&1 | grep -v amdgpu.ids > $O/segment_timeline.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_roof -o roof -- python3 $R/bench.py --roofline-only > $O/roofline_only.json 2> $O/prof_roof.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o train -- python3 $R/bench.py --workload train --no-cpu-baseline --no-roofline-top --no-parity --steps 50 > $O/prof_train.json 2> $O/prof_train.log
echo prof_train done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fwd -o fwd -- python3 $R/bench.py --workload fwd --no-cpu-baseline --no-parity --steps 50 > $O/prof_fwd.json 2> $O/prof_fwd.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hires_train -o hires_train -- python3 $R/bench.py --workload hires-train --steps 10 > $O/prof_hires_train.json 2> $O/prof_hires_train.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hires_train_da -o hires_train_da -- python3 $R/bench.py --workload hires-train --da res,decoders --steps 10 > $O/prof_hires_train_da.json 2> $O/prof_hires_train_da.log
python3 $R/profiles/step_timeline.py $O/prof_train > $O/step_timeline.txt 2>&1
python3 $R/profiles/fwd_timeline.py $O/prof_fwd > $O/fwd_timeline.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c --output-format csv -- python3 $R/profiles/pmc_resconv.py > $O/pmc_$c.log 2>&1
done
python3 - <<PY > $O/pmc_resconv.json
import csv, glob, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$O/pmc_%s/**/*counter_collection.csv" % c, recursive=True)
    v = sorted(float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "resconv_kernel" in r["Kernel_Name"])
    out[c + "_KB_median"] = v[len(v) // 2]; out["launches_sampled"] = len(v)
print(json.dumps(out))
PY
# keep what travels back small: statistics only, no traces
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -delete
ls $O
