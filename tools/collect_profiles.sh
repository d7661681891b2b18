# usage (on the GPU box, from the repo root): bash tools/collect_profiles.sh <tag>
# kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes, of the default bench command; summaries -> profiles/
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$tag && mkdir -p $R/gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag/stats -o s -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/prof_$tag/bench_under_rocprof.json 2> $R/gpurun_out/prof_$tag/stats.err || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_$tag/fetch -o f -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof_$tag/fetch.err || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_$tag/write -o w -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof_$tag/write.err || exit 1
cd $R && python tools/make_profile_summary.py $tag gpurun_out/prof_$tag/stats gpurun_out/prof_$tag/fetch gpurun_out/prof_$tag/write > gpurun_out/prof_$tag/summary.txt
grep "^{" gpurun_out/prof_$tag/bench_under_rocprof.json > profiles/${tag}_bench_under_rocprof.json
cp profiles/${tag}_kernel_stats.csv profiles/${tag}_pmc.json profiles/pmc_traffic.json profiles/${tag}_bench_under_rocprof.json gpurun_out/prof_$tag/
rm -rf gpurun_out/prof_$tag/stats gpurun_out/prof_$tag/fetch gpurun_out/prof_$tag/write
cat gpurun_out/prof_$tag/summary.txt | head -60
