#!/bin/bash
# Diagnostic: tile dispatch order (plan._tiles) at several batch sizes: plain XCD order against short workgroups last.
run() {  # label args...
  local label=$1; shift
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-through-api "$@" 2>/dev/null | python3 -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('$label', os.environ.get('WB_TILE_ORDER'), '%.4g' % d['value'], round(d['ms_per_step'],5))"
}
for b in 2 4 8 16; do
for o in natural short natural short; do
export WB_TILE_ORDER=$o; run "B=$b" --batch $b --steps 40 --warmup 4 --pool 2 --streams 2
done; done
