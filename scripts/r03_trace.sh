#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03tr}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rk in ${RANKS:-4}; do
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace$rk -o solo -- python $R/scripts/mgpu_solo_trace.py $rk 30 > $O/solo$rk.log 2>&1
grep "period" $O/solo$rk.log
f=$(ls $O/trace$rk/*/*kernel_trace.csv $O/trace$rk/*kernel_trace.csv 2>/dev/null | head -1)
python $R/scripts/trace_timeline.py $f 6 > $O/timeline$rk.txt 2>&1; cat $O/timeline$rk.txt
head -1 $f > $O/header$rk.txt
rm -rf $O/trace$rk
done
