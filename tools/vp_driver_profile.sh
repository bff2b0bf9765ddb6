#!/bin/bash
# host/rbphdslam_vp on the committed 900-message Victoria Park extract, 5000 particles: three untraced processes of three passes each, without log files (wall of every pass: the first carries the GPU's clock ramp)
# and one rocprofv3 --kernel-trace --stats run (kernel time).   usage: tools/vp_driver_profile.sh <tag>   (GPU box)
set -u
TAG=${1:-rXX}
OUT=gpurun_out/vpdrv_$TAG
mkdir -p $OUT
ROOT="$GRAFT_REPO_ROOT"
CMD="$ROOT/rfs-slam_amd/host/rbphdslam_vp -c $ROOT/tests/golden/rbphdslam_VictoriaPark_c4.xml -d $ROOT/tests/golden/vp_extract -n 5000 -s 3 --repeat 3 ${VPD_ARGS:-}"
for k in 1 2 3; do $CMD > $OUT/run$k.log 2>&1; grep -h "^particles" $OUT/run$k.log; done
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
grep -h "^particles" $OUT/trace.log
python - <<PY
import csv, glob
tot = 0.0; rows = []
for f in glob.glob("$OUT/trace/**/trace_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((r["Name"].split("(")[0][:60], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6))
        tot += float(r["TotalDurationNs"]) / 1e6
for n, c, t in sorted(rows, key=lambda x: -x[2])[:10]:
    print("%-60s %6d %9.2f ms" % (n, c, t))
print("kernel time total %.1f ms" % tot)
PY
