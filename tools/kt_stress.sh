set -e
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp
rm -rf $O/kt_stress
rocprofv3 --kernel-trace --output-format csv -d $O/kt_stress -- python3 $R/bench.py --stress --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-table > $O/kt_stress.log 2>&1
python3 $R/tools/prof_summary.py $O/kt_stress 25 > $O/kt_stress.txt
rm -rf $O/kt_stress
cat $O/kt_stress.txt
