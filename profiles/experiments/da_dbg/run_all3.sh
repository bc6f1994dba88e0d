#!/bin/bash
D=profiles/experiments/da_dbg
O=gpurun_out/da_dbg; mkdir -p $O
run() { echo "=== $*" | tee -a $O/log3.txt; timeout -k 10 300 python $D/run_da_dbg.py "$@" >> $O/log3.txt 2>&1; echo "exit $?" >> $O/log3.txt; }
run $D/lib_exp1.so --region 0 --neighbour wide --dbg --runs 8
run $D/lib_exp2.so --region 0 --neighbour wide --dbg --runs 8
run $D/lib_exp4.so --region 0 --neighbour wide --dbg --runs 8
run $D/lib_vgprform.so --region 0 --neighbour wide --dbg --runs 8
grep -v "^       [WR] {" $O/log3.txt | grep "===\|RESULT\|W summary\|hashes" | tail -n 70
