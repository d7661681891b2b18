cd /tmp && export TMPDIR=/tmp
for sk in 1 2; do
  export ZR_DEBUG_SKIP=$sk
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_sk$sk -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/trace_sk$sk.log 2>&1 || exit 1
  (cd $GRAFT_REPO_ROOT && echo "== skip $sk" && python tools/trace_tail.py gpurun_out/trace_sk$sk 40 | grep -E "raster|cull|resolve|sum") 
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace_sk$sk
done
