# usage: bash tools/run_trace.sh <tag> [bench args...]   -> gpurun_out/trace_<tag>_seq.txt (per-dispatch means of the last frames)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_$tag -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/trace_$tag.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT && python tools/trace_tail.py gpurun_out/trace_$tag 40 > gpurun_out/trace_${tag}_seq.txt
rm -rf gpurun_out/trace_$tag
cat gpurun_out/trace_${tag}_seq.txt
