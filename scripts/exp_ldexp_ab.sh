#!/bin/bash
# rt_exp ending with v_ldexp_f32 on the device (round 4): exhaustive bit check of the two endings, parity tests that exercise the filters, then the A/B of the bench line.
# usage (gpurun): bash scripts/exp_ldexp_ab.sh <tag>
R=$GRAFT_REPO_ROOT; T=${1:-r04exp}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -I include scripts/probe/exp_ldexp_probe.hip -o /tmp/exp_probe 2>/dev/null && /tmp/exp_probe | tee $O/probe.txt
timeout 900 python -m pytest tests -q -m gpu -x -k "denoise or filter or detmath or golden or all_stages" 2>&1 | tail -2 | tee $O/tests.txt
# (the A/B ran with the ldexp ending applied to include/rt_detmath.h behind RT_EXP_NO_LDEXP; not kept, see profiles/r04_exp_ldexp_ab.txt)
bash scripts/variants_bench.sh $T "ldexp|-|-" "multiplies|-DRT_EXP_NO_LDEXP|-" "ldexp_again|-|-" | tee $O/ab.txt
