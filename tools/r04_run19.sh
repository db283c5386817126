#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for i in 1 2; do
python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids
EIGSOLVE_S2_PRIO_EXPERIMENT=1 python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids | tail -2
EIGSOLVE_S2_PRIO_EXPERIMENT=0 python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids | tail -2
done
