#!/bin/bash
# A/B of conv planner settings: matting net per layer + human-parsing net (batch 4, planned for 4) per shape.
out=gpurun_out/conv_ab; mkdir -p $out
run() {  # name, env...
  name=$1; shift
  env "$@" python tools/dev/profile_convs.py > $out/unet_$name.txt 2>&1 || return 1
  env "$@" B=4 PLAN=4 python tools/dev/profile_schp.py > $out/schp_$name.txt 2>&1 || return 1
  echo "$name: $(grep 'whole forward' $out/unet_$name.txt) | $(tail -1 $out/schp_$name.txt)"
}
run rg4 USC_RG_MIN=4 && run rg2 USC_RG_MIN=2 && run rg1 USC_RG_MIN=1
