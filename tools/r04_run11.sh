#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids
for nb in 32 48; do EIGSOLVE_TRD_NB=$nb python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids; done
for b in 128 512; do EIGSOLVE_BT_NB=$b python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids; done
EIGSOLVE_GST_THR=512 python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids
EIGSOLVE_GST_THR=2048 python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids
EIGSOLVE_GST=3 EIGSOLVE_TRSM_BASE=512 python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids
EIGSOLVE_TRSM_BASE=512 python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids
EIGSOLVE_OVERLAP=0 python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids
echo "--- C2"
python tools/iso_phases.py 2048 512 real 2>&1 | grep -v amdgpu.ids
for nb in 32 48; do EIGSOLVE_TRD_NB=$nb python tools/iso_phases.py 2048 512 real 2>&1 | grep -v amdgpu.ids; done
EIGSOLVE_TRD_FINISH=32 python tools/iso_phases.py 2048 512 real 2>&1 | grep -v amdgpu.ids
