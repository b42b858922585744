#!/bin/bash
cd "$GRAFT_REPO_ROOT"
p() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['ms_per_step'],4), round(d['value']), d.get('roofline',{}).get('avg_us'))"; }
for w in 0 5 10 15 20 30 40 60 100; do
python3 bench.py --cpu-seconds 0 --steps 5 --warmup $w --no-extras --no-kernel-timers 2>/dev/null | p s5w$w
done
