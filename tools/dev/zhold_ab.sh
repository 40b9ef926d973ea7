#!/bin/bash
# dev: k_ibp_ztile with the pre-update state held in registers (default, two tiles per CU) against re-read (SRX_ZTILE_HOLD=0, three per CU), one box
for lib in default "$@" default; do
    if [ "$lib" = default ]; then unset SRX_LIB; else export SRX_LIB=$lib; fi
    for wl in "c3_mono 1" "c3_mono 8" "c3_mono_measured 1"; do
        set -- $wl
        python3 bench.py --workload $1 --batch $2 --no-cpu-baseline --no-secondary --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${lib##*/}', '$1 x$2', d['config']['path'], 'ms/step', d['ms_per_step'], 'iter us', d['roofline']['iteration_kernels_us'], 'frac', d['roofline']['frac'])"
    done
done
