for lib in base lean1 noatom; do
  for flag in "" "--untextured"; do
    R3N_LIB=$PWD/variants/lib_$lib.so python bench.py --steps 100 --warmup 10 --no-cpu-baseline $flag 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$lib','$flag',d['ms_per_step'],{k:round(v*1e3,1) for k,v in d['stage_ms_per_frame'].items() if v})"
  done
done
