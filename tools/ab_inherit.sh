cd $GRAFT_REPO_ROOT
for mode in reference eager; do
for ts in "1 1" "1 2" "2 1" "2 2" "3 1" "3 2" "4 1" "5 3"; do
set -- $ts
RFSGPU_BIRTH_INHERITANCE=$mode ./rfs-slam_amd/host/rbphdslam2d_sim -c tests/golden/rbphdslam2dSim_c1.xml -t $1 -s $2 -n 200 | grep RESULT | sed "s/^/$mode t=$1 s=$2 /"
done; done
