cd $GRAFT_REPO_ROOT
for V in "512 640" "2000 640" "4096 640" "2000 2000" "4096 2000"; do
  set -- $V
  RESTIR_LAT_TILES_SHARED=$1 RESTIR_LAT_TILES_IND_SHARED=$2 timeout 1200 python bench.py --emulate-world 8 --steps 30 --warmup 12 2>/dev/null | python -c "
import json,sys,statistics
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['rank_period_ms']
print('shared thresholds $1 $2: period slowest', d['slowest_rank_period_ms'], 'median', round(statistics.median(p),3), p)"
done
