#!/bin/bash
D=profiles/experiments/da_dbg
O=gpurun_out/da_dbg; mkdir -p $O
run() { echo "=== $*" | tee -a $O/log2.txt; timeout -k 10 300 python $D/run_da_dbg.py "$@" >> $O/log2.txt 2>&1; echo "exit $?" >> $O/log2.txt; }
run $D/lib_dbgw.so --region 0 --neighbour wide --dbg --runs 10
run $D/lib_dbgo.so --region 0 --neighbour wide --dbg --runs 10
run $D/lib_dbgr.so --region 0 --neighbour wide --dbg --runs 10
grep -v "^       [WR] {" $O/log2.txt | tail -n 70
