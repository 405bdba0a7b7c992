#!/bin/bash
# Same-box A/B of library builds / env settings on the bench scene.  usage (via gpurun): bash scripts/ab_libs.sh "<label>|<env assignments>" ...
# prints pipelined ms/frame, serial sum and the serial stage times per variant
for v in "$@"; do
  label="${v%%|*}"; envs="${v#*|}"
  env $envs python bench.py --no-cpu-baseline > /tmp/ab_$$.json 2>/dev/null
  python - "$label" /tmp/ab_$$.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
s = d["roofline"]["serial"]["stage_ms_per_frame"]
print("%-28s pipelined %.3f  sustained %.3f  serial %.3f | direct %.3f indirect %.3f filters %.3f + %.3f" % (sys.argv[1], d["ms_per_step"], d["sustained"]["ms_per_frame"], d["ms_per_frame_serial"],
      s["direct_stage"], s["indirect_stage"], s["denoise_direct"], s["denoise_indirect"]), flush=True)
PY
done
