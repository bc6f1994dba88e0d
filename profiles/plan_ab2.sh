# A/B: workgroup target of the InstanceNorm backward's spatial split (pointwise.hip hdrsky_norm_act_bwd_nslices, HDRSKY_NAB_TARGET)
run() { env $1 python bench.py --workload train --no-cpu-baseline --no-roofline-top --no-parity --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s' % '$1', d['ms_per_step'])"; }
for rep in 1 2; do
run "HDRSKY_X=default"
run "HDRSKY_NAB_TARGET=256"
run "HDRSKY_NAB_TARGET=128"
run "HDRSKY_NAB_TARGET=64"
run "HDRSKY_NAB_TARGET=1024"
done
