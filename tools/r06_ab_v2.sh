#!/bin/bash
# GPU box: A/B of two variant libraries on the --bistro-v2 line at EQUAL step counts (the stage table is taken where the camera
# stands after the timed steps, and the foliage in view changes along the dolly).   usage: tools/r06_ab_v2.sh A B
for st in 20 40 60; do
  echo "== steps $st"
  for v in "$@"; do
    R3N_LIB=$PWD/variants/lib_$v.so python bench.py --bistro-v2 --no-cpu-baseline --steps $st --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_frame']
print('$v', d['ms_per_step'], 'raster_big_cut', round(s['raster_big_cut']*1e3,1), 'shadow_raster_big', round(s['shadow_raster_big']*1e3,1), 'raster_cut', round(s['raster_cut']*1e3,1), 'shadow_raster', round(s['shadow_raster']*1e3,1))"
  done
done
