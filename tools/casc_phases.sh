#!/bin/bash
# cumulative cascade kernel time when the kernel returns after a phase (WB_CASC_DBG), binned vs planar float tile
# usage: tools/casc_phases.sh [batch]   (prints cascade_tile_ms per variant)
B=${1:-32}
for nb in 0 1; do
  for dbg in 2 4 8 16 32 0; do
    if [ $nb = 1 ]; then export WB_CASC_NOBIN=1; else unset WB_CASC_NOBIN; fi
    WB_CASC_DBG=$dbg python bench.py --batch $B --steps 5 --warmup 1 --repeats 2 --pool 1 --streams 1 --no-cpu-baseline --no-through-api --stages 127 2>/dev/null \
      | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('nobin=$nb dbg=$dbg cascade_tile_ms=%.4f per_image_us=%.2f' % (j['kernels']['cascade_tile_ms'], j['kernels']['cascade_tile_ms']*1e3/$B))"
  done
done
