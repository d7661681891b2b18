# usage: bash tools/run_timeline.sh <tag> [bench args...]   -> gpurun_out/timeline_<tag>.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_$tag -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras "$@" > $GRAFT_REPO_ROOT/gpurun_out/tl_$tag.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT && python tools/timeline.py gpurun_out/tl_$tag 10 > gpurun_out/timeline_$tag.txt
rm -rf gpurun_out/tl_$tag
