#!/bin/bash
# Tuning aid (GPU box): the UNMODIFIED reference 2-D driver (tests/support/_build/rbphdslam2dSim, built where /root/reference exists) at 2000 particles, 600 steps,
# result logging off, with the binding's own breakdown (RFSGPU_BINDING_PROFILE=1): TimingInfo totals + where predict() and the resample tail spend their host time.
# Environment passes through (RFSGPU_LAZY_PREDICT=0, RFSGPU_RESAMPLE_SYNC=1 for the A/B).
set -e
T=$(mktemp -d)
python - <<PY
import re
s=open('tests/golden/rbphdslam2dSim_c1.xml').read()
s=s.replace("<config>","<config>\n  <logging><logResultsToFile>0</logResultsToFile><logTimingToFile>0</logTimingToFile><logDirPrefix>$T/</logDirPrefix></logging>",1)
s=re.sub(r"<timesteps>\d+</timesteps>","<timesteps>600</timesteps>",s)
s=re.sub(r"<nParticles>\d+</nParticles>","<nParticles>2000</nParticles>",s)
open("$T/cfg.xml","w").write(s)
PY
RFSGPU_GM_CAPACITY=256 RFSGPU_BINDING_PROFILE=1 tests/support/_build/rbphdslam2dSim -c $T/cfg.xml -t 1 -s 1 2>&1 | grep -i "rfsgpu binding\|Resampling\|Prediction\|Map Update\|Total"
