# rocprofv3 kernel trace + stats of the volume build alone.  Usage: gpurun -- bash tools/tools_prof_volume.sh TAG [cfg2]
TAG=${1:-volprof}; CFG=${2:-cfg2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/tools_volume_bench.py $CFG 20 > $OUT/run.log 2>&1
find $OUT -type f ! -name "*kernel_stats*" ! -name "*.log" -delete
find $OUT -name "*kernel_stats*" | head -1 | xargs cat | cut -d, -f1-7 | head -12
