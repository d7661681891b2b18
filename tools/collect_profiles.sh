# usage (on the GPU box, from the repo root): bash tools/collect_profiles.sh <tag> [extra bench args, e.g. ZR_SERIAL via env]
# kernel-trace stats, then FETCH_SIZE, WRITE_SIZE and SQ_INSTS_VALU in separate --pmc passes, of the default bench workload
# (no extras: one timed loop only); summaries -> profiles/<tag>_*
tag=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rm -rf $O && mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras "$@" > $O/bench_under_rocprof.json 2> $O/stats.err || { tail -5 $O/stats.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" > /dev/null 2> $O/fetch.err || { tail -5 $O/fetch.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" > /dev/null 2> $O/write.err || { tail -5 $O/write.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/sq -o q -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" > /dev/null 2> $O/sq.err || { tail -5 $O/sq.err; exit 1; }
cd $R && python tools/make_profile_summary.py $tag $O/stats $O/fetch $O/write $O/sq $O/bench_under_rocprof.json > $O/summary.txt || { tail -5 $O/summary.txt; exit 1; }
grep "^{" $O/bench_under_rocprof.json > profiles/${tag}_bench_under_rocprof.json
cp profiles/${tag}_kernel_stats.csv profiles/${tag}_pmc.json profiles/current.json profiles/${tag}_bench_under_rocprof.json $O/
rm -rf $O/stats $O/fetch $O/write $O/sq
head -60 $O/summary.txt
