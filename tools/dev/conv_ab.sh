#!/bin/bash
# A/B of the conv kernel variants / heuristics: matting net per layer + human-parsing net (batch 4) per shape.
out=gpurun_out/conv_ab; mkdir -p $out
run() {  # name, env...
  name=$1; shift
  env "$@" python tools/dev/profile_convs.py > $out/unet_$name.txt 2>&1 || return 1
  env "$@" B=4 python tools/dev/profile_schp.py > $out/schp_$name.txt 2>&1 || return 1
  echo "$name: $(grep 'whole forward' $out/unet_$name.txt) | $(tail -1 $out/schp_$name.txt)"
}
run kw9 USC_KW_STEPS=9 PLAN=4 && run kw2 PLAN=4 && run kw4 USC_KW_STEPS=4 PLAN=4 && run ms6 USC_MIN_STEPS=6 PLAN=4 && run ms12 USC_MIN_STEPS=12 PLAN=4
