"""Headline fields of bench.py's JSON line from stdin: python bench.py ... | python tools/bench_ms.py"""
import json,sys
for l in sys.stdin:
    if l.startswith('{"metric"'):
        j=json.loads(l); print("ms_per_step", round(j["ms_per_step"],4), "steps", j["steps"], "warmup", j["warmup"], "fwd_only", j.get("forward_only",{}).get("value"))
