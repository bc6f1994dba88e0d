# A/B: the step as ONE hipGraph (Trainer.capture(whole=True)) against one graph per segment, same box, three rounds
run() { env $1 python bench.py --workload train --no-cpu-baseline --no-roofline-top --no-parity --steps 100 --warmup 10 2>gpurun_out/ab_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s' % '$1', d['ms_per_step'])"; }
for rep in 1 2 3; do
run "HDRSKY_X=default"
run "HDRSKY_WHOLE_GRAPH=0"
done
tail -5 gpurun_out/ab_err.txt
