#!/bin/bash
# experiment: spectrogram kernel variants (store pattern) timed with the audio bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/dali_amd/csrc
for V in 0; do
  touch audio.hip; make CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I../../include -Wno-unused-function -DSPEC_EXP=$V" > /dev/null 2>&1
  echo "variant $V"
  (cd $R && python bench.py --full-line --workload audio --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['operator_device_ms'])")
done
