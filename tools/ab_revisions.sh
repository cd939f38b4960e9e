#!/bin/bash
# R&D: the bench line of this tree against the tree of another revision exported under tools/variants/old (its own
# liblbmi.so built there), alternating on the same box:  bash tools/ab_revisions.sh [bench args]
for round in 1 2 3; do
  for tree in . tools/variants/old; do
    (cd $tree && timeout -k 10 200 python bench.py --cpu-baseline 0 --steps 100 "$@" 2>/dev/null) | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline') or {}; e = d.get('hydro_every_step') or {}
        print('%-22s %8.1f MLUPS  kernel %.5f ms  every-step %s MLUPS (%s ms)' % ('$tree', d['value'], r.get('avg_launch_ms', 0), e.get('value'), e.get('avg_launch_ms')))
"
  done
done
