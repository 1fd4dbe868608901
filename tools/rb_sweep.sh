#!/bin/bash
# A/B sweep of residual-block kernel variants in one box session (clock drifts between sessions: compare only within one).
# usage: rb_sweep.sh "RB TM RING" ...   (RB: 3 = k_resblock3, 1 = k_resblock; TM 0 = automatic)
mkdir -p gpurun_out/st
cfgs=("$@")
for rep in 1 2; do
  for cfg in "${cfgs[@]}"; do
    read -r rb tm ring <<< "$cfg"
    f=gpurun_out/st/sw_${rb}_${tm}_${ring}.bin
    out=$(GAZ_RB=$rb GAZ_RB_TM=$tm GAZ_RB_RING=$ring GAZ_RB_STAMPS=$f timeout -k 10 120 python tools/eval_probe.py --repeats 20 | tail -1)
    st=$(python tools/rb_stamps.py $f | grep -E "kernel span|shader clock" | tr '\n' ' ')
    echo "RB=$rb TM=$tm RING=$ring | $out | $st"
  done
done
