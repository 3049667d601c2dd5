#!/bin/bash
# development aid (GPU box): parity tests of the sample kernels, phase stamps, and the timed sample kernel
mkdir -p gpurun_out
export DSS_LPCNET_SYNTHETIC=1
python -m pytest tests/test_gpu_lpcnet.py -x -q -m gpu 2>&1 | tail -2 &&
python tools/phase_stamps.py 256 > gpurun_out/stamps.txt 2>&1 && cat gpurun_out/stamps.txt &&
python tools/ab_time.py tools/ab/base.so delayed-speech-synthesis_amd/libdss_hip.so 2>&1 | tail -6
