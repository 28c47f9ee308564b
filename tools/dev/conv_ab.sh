#!/bin/bash
# A/B of the conv kernel variants / heuristics: matting net per layer + human-parsing net (batch 4) per shape.
out=gpurun_out/conv_ab; mkdir -p $out
run() {  # name, env...
  name=$1; shift
  env "$@" python tools/dev/profile_convs.py > $out/unet_$name.txt 2>&1 || return 1
  env "$@" B=4 python tools/dev/profile_schp.py > $out/schp_$name.txt 2>&1 || return 1
  echo "$name: $(grep 'whole forward' $out/unet_$name.txt) | $(tail -1 $out/schp_$name.txt)"
}
run base USC_MIN_STEPS=9 && run q4 USC_MIN_STEPS=9 USC_F_TARGET=256 USC_WG_MIN=64 USC_KW_TARGET=192 && run q3 USC_MIN_STEPS=9 USC_F_TARGET=384 USC_WG_MIN=64 USC_KW_TARGET=256 && run q2 USC_MIN_STEPS=9 USC_F_TARGET=512 USC_WG_MIN=128 USC_KW_TARGET=384 && run q4b USC_MIN_STEPS=9 USC_F_TARGET=256 USC_WG_MIN=32 USC_KW_TARGET=192
