#!/bin/bash
# Development aid: build the working tree's library as tools/ab/<name>.so with extra compiler flags (for tools/ab_time.py).
set -e
name=$1; shift
mkdir -p tools/ab
PYTHONPATH=delayed-speech-synthesis_amd python3 - "$name" "$@" <<'PY'
import os, sys
from dss_amd import build
out = os.path.abspath(os.path.join("tools", "ab", sys.argv[1] + ".so"))
print("built", build.build_library(extra_flags=sys.argv[2:], out=out))
PY
