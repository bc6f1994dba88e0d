#!/bin/bash
# one gpurun call: every diagnostic variant in turn (own process each), log to gpurun_out/da_dbg/
D=profiles/experiments/da_dbg
O=gpurun_out/da_dbg; mkdir -p $O
run() { echo "=== $*" | tee -a $O/log.txt; timeout -k 10 240 python $D/run_da_dbg.py "$@" >> $O/log.txt 2>&1; echo "exit $?" >> $O/log.txt; }
run $D/lib_base.so --region 0 --neighbour wide
run $D/lib_base.so --region 1 --neighbour wide
run $D/lib_dbg.so --region 0 --neighbour wide --dbg --runs 4
run $D/lib_pad.so --region 0 --neighbour wide
run $D/lib_pad.so --region 1 --neighbour wide
run $D/lib_noslp.so --region 0 --neighbour wide
run $D/lib_base.so --region 0 --neighbour gemm
run $D/lib_base.so --region 0 --neighbour conv
run $D/lib_base.so --region 0 --neighbour stream
run $D/lib_base.so --region 0 --neighbour wide --shape lowres
tail -n 80 $O/log.txt
