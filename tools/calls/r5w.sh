cd $GRAFT_REPO_ROOT
O=gpurun_out/r5w; mkdir -p $O
for i in $(seq 1 24); do timeout 120 python tools/first_launch_probe.py $i 2>&1 | grep "seed\|Error\|error" ; done > $O/first_launch.txt; cat $O/first_launch.txt
