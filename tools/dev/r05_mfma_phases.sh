#!/bin/bash
# timing builds of the library (conv_mfma_kernel stamps; with / without the raised prologue / epilogue priority) + tools/dev/mfma_phases.py,
# then the un-instrumented per-layer times of the parsing net (12 frames) and the matting net with both priority settings
set -e
R=$PWD
mkdir -p gpurun_out/r5_mfma
make -s -j8 -C video_unscreen_amd/csrc
make -s -j8 -C video_unscreen_amd/csrc EXTRA=-DUSC_MFMA_TIMING OBJDIR=build_timing OUT=../libunscreen_hip_timing.so
make -s -j8 -C video_unscreen_amd/csrc EXTRA="-DUSC_MFMA_TIMING -DUSC_MFMA_NO_PRIO" OBJDIR=build_timing_noprio OUT=../libunscreen_hip_timing_noprio.so
make -s -j8 -C video_unscreen_amd/csrc EXTRA=-DUSC_MFMA_NO_PRIO OBJDIR=build_noprio OUT=../libunscreen_hip_noprio.so
USC_TIMING_LIB=$R/video_unscreen_amd/libunscreen_hip_timing.so python tools/dev/mfma_phases.py "$@" 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_mfma/phases_prio.txt
USC_TIMING_LIB=$R/video_unscreen_amd/libunscreen_hip_timing_noprio.so python tools/dev/mfma_phases.py "$@" 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_mfma/phases_noprio.txt
grep -h "==\|tables\|epilogue\|K loop  \|lifetime" gpurun_out/r5_mfma/phases_prio.txt | cut -c1-200
python tools/dev/ab_nets.py video_unscreen_amd/libunscreen_hip.so video_unscreen_amd/libunscreen_hip_noprio.so video_unscreen_amd/libunscreen_hip.so video_unscreen_amd/libunscreen_hip_noprio.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_mfma/ab_nets.txt
