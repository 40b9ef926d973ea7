#!/bin/bash
# dev: window shapes of k_ibp_bfwd / k_ibp_bbwd on one box -- one rgb_cal_target-shaped frame and eight, Gaussian and measured PSF
#   (builds: hipcc ... -DSRX_BT_NBY=.. -DSRX_BT_NBX=.. -DSRX_BT_MINB=.. -> enph459-super-resolution_amd/build/libsrx_bt<shape>.so)
for lib in default "$@"; do
    if [ "$lib" = default ]; then unset SRX_LIB; else export SRX_LIB=$lib; fi
    for wl in "c3_rgb 1" "c3_rgb 8" "c3_rgb_measured 1"; do
        set -- $wl
        for rep in 1 2; do
        python3 bench.py --workload $1 --batch $2 --no-cpu-baseline --no-secondary --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${lib##*/}', '$1 x$2', d['config']['path'], 'ms/step', d['ms_per_step'], 'iter us', d['roofline']['iteration_kernels_us'], 'frac', d['roofline']['frac'])"
        done
    done
done
