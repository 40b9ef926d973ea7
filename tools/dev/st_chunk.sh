#!/bin/bash
# dev: C2 in float64 through the strip kernels, batch chunks of different sizes (does the chunk's working set stay in the Infinity Cache?)
for c in ${CHUNKS:-0 128}; do
  SRX_STILE_CHUNK=$c timeout -k 10 200 python3 bench.py --precision f64 --no-cpu-baseline --no-secondary --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('chunk $c', d['config']['path'], 'ms/step', d['ms_per_step'], {k:v['total_ms'] for k,v in (d.get('kernels') or {}).items()}, (d.get('roofline') or {}).get('frac'))" || exit 1
done 2>&1 | tee gpurun_out/st_chunk.log
