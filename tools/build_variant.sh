#!/bin/bash
# Development aid: build the working tree's library as tools/ab/<name>.so with extra compiler flags (for tools/ab_time.py).
set -e
name=$1; shift
mkdir -p tools/ab
cd delayed-speech-synthesis_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -x hip \
    -Wno-unused-result -Wno-unused-value "$@" -o ../../tools/ab/$name.so \
    dss_capi.cpp hga_kernels.hip lpcnet_frame.hip lpcnet_sample.hip lpcnet_sample_pair.hip lpcnet_sample_generic.hip speech_gate.hip vad_lstm.hip bilstm_decoder.hip
echo built tools/ab/$name.so
