#!/bin/bash
# cumulative kernel time when the kernel returns after a phase (WB_CASC_DBG / WB_CHAN_DBG), per image
# usage: tools/casc_phases.sh [batch]
B=${1:-32}
run() { python bench.py --batch $B --steps 5 --warmup 1 --repeats 2 --pool 1 --streams 1 --no-cpu-baseline --no-through-api --stages 127 2>/dev/null; }
for dbg in 2 4 8 16 32 0; do
  WB_CASC_DBG=$dbg run | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('casc dbg=$dbg cascade_tile per_image_us=%.2f' % (j['kernels']['cascade_tile_ms']*1e3/$B))"
done
for dbg in 1 2 4 0; do
  WB_CHAN_DBG=$dbg run | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('chan dbg=$dbg channels per_image_us=%.2f' % (j['kernels']['channels_ms']*1e3/$B))"
done
WB_NO_RANKS=1 run | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('float channels + planar tile: channels %.2f cascade_tile %.2f us per image' % (j['kernels']['channels_ms']*1e3/$B, j['kernels']['cascade_tile_ms']*1e3/$B))"
