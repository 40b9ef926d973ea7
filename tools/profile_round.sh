#!/bin/bash
# profile_round.sh TAG -- the rocprofv3 evidence of one round, collected on the GPU box in ONE gpurun call:
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/profile_round.sh r03 "c2 c3_mono"'
# For each workload named (default: every workload bench.py reports a roofline for):
#   <wl>_stats : rocprofv3 --kernel-trace --stats            (per-kernel durations)
#   <wl>_fetch : rocprofv3 --kernel-trace --pmc FETCH_SIZE   (separate pass, as MI355X_MICROARCH.md prescribes)
#   <wl>_write : rocprofv3 --kernel-trace --pmc WRITE_SIZE
#   <wl>_sq1/2 : SQ / GRBM counters (VALU and LDS activity, wave-cycle split)
# Everything lands in gpurun_out/TAG/; tools/profile_collect.py TAG turns it into profiles/TAG_* afterwards (on the dev box).
# The program after "--" is python3 itself (no env / bash -c hop: the profiler's preload initialises the GPU first).
set -e
TAG=${1:-r03}
WLS=${2:-"c2 c3_mono c3_mono_measured c3_mono@f64 c3_f4 c3_rgb c3_rgb_measured"}   # name[@precision]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
COMMON="--steps 1 --no-cpu-baseline --no-roofline --no-secondary"
SQ1="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
SQ2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
for wlp in $WLS; do
    wl=${wlp%@*}; prec=f32; [ "$wlp" != "$wl" ] && prec=${wlp#*@}
    tag=$wl; [ "$prec" != f32 ] && tag=${wl}_$prec
    echo "== $wl ($prec) kernel stats"
    # (six calls: the first call after a pause runs the long kernels ~8 % slow, which a two-call average would show)
    timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${tag}_stats" -o run -- python3 "$R/bench.py" --workload $wl --precision $prec --warmup 2 --steps 4 --no-cpu-baseline --no-roofline --no-secondary > "$O/${tag}_stats.log" 2>&1
    echo "== $wl FETCH_SIZE"
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/${tag}_fetch" -o run -- python3 "$R/bench.py" --workload $wl --precision $prec --warmup 0 $COMMON > "$O/${tag}_fetch.log" 2>&1
    echo "== $wl WRITE_SIZE"
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/${tag}_write" -o run -- python3 "$R/bench.py" --workload $wl --precision $prec --warmup 0 $COMMON > "$O/${tag}_write.log" 2>&1
    echo "== $wl SQ pass 1"
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d "$O/${tag}_sq1" -o run -- python3 "$R/bench.py" --workload $wl --precision $prec --warmup 0 $COMMON > "$O/${tag}_sq1.log" 2>&1
    echo "== $wl SQ pass 2"
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d "$O/${tag}_sq2" -o run -- python3 "$R/bench.py" --workload $wl --precision $prec --warmup 0 $COMMON > "$O/${tag}_sq2.log" 2>&1
    echo "== $wl TCC request sizes (FETCH_SIZE's own terms: 32 / 64 / 128-byte fabric reads; diagnostic, may fail without harm)"
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_BUBBLE_sum --output-format csv -d "$O/${tag}_tcc" -o run -- python3 "$R/bench.py" --workload $wl --precision $prec --warmup 0 $COMMON > "$O/${tag}_tcc.log" 2>&1 || echo "   (TCC pass failed)"
done
echo "== bench lines"
cd "$R"
for wlp in $WLS; do
    wl=${wlp%@*}; prec=f32; [ "$wlp" != "$wl" ] && prec=${wlp#*@}
    tag=$wl; [ "$prec" != f32 ] && tag=${wl}_$prec
    if [ "$tag" = c2 ]; then
        timeout -k 10 600 python3 bench.py --steps 3 --warmup 1 > "$O/c2_bench.json" 2> "$O/c2_bench.err"
    else
        timeout -k 10 200 python3 bench.py --workload $wl --precision $prec --steps 3 --warmup 1 --no-cpu-baseline > "$O/${tag}_bench.json" 2> "$O/${tag}_bench.err"
    fi
done
tail -c 600 "$O/c2_bench.json" || true
echo PROFILE_ROUND_DONE
