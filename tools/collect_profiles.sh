# usage (on the GPU box, from the repo root): bash tools/collect_profiles.sh <tag> [extra bench args, e.g. --serial / --config 4]
# rocprofv3 --kernel-trace --stats of the bench workload, then the counters in SEPARATE --pmc passes (never together with a trace):
#   FETCH_SIZE | WRITE_SIZE | SQ issue/wait set | SQ LDS/memory set        -> profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json
# and the same SQ issue set over tools/valu_calib (what a saturated SIMD reads on these counters)  -> profiles/<tag>_valu_calib*.json
tag=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
SQ_A="SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
SQ_B="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA"
SQ_C="SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM"      # lanes active per vector instruction = THREAD_CYCLES_VALU / (64 x ACTIVE_INST_VALU)
cd /tmp && export TMPDIR=/tmp
rm -rf $O && mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --no-extras"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $B --steps 100 --warmup 10 "$@" > $O/bench_under_rocprof.json 2> $O/stats.err || { tail -5 $O/stats.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- $B --steps 20 --warmup 5 "$@" > /dev/null 2> $O/fetch.err || { tail -5 $O/fetch.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- $B --steps 20 --warmup 5 "$@" > /dev/null 2> $O/write.err || { tail -5 $O/write.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc $SQ_A --output-format csv -d $O/sqa -o q -- $B --steps 20 --warmup 5 "$@" > /dev/null 2> $O/sqa.err || { tail -5 $O/sqa.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc $SQ_B --output-format csv -d $O/sqb -o q -- $B --steps 20 --warmup 5 "$@" > /dev/null 2> $O/sqb.err || { tail -5 $O/sqb.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc $SQ_C --output-format csv -d $O/sqc -o q -- $B --steps 20 --warmup 5 "$@" > /dev/null 2> $O/sqc.err || { tail -5 $O/sqc.err; exit 1; }
if [ -x $R/tools/valu_calib ]; then
  $R/tools/valu_calib > $R/profiles/${tag}_valu_calib.json 2> $O/calib.err || tail -3 $O/calib.err
  timeout -k 10 300 rocprofv3 --pmc $SQ_A --output-format csv -d $O/calib -o c -- $R/tools/valu_calib > /dev/null 2>> $O/calib.err || tail -3 $O/calib.err
fi
cd $R && python tools/make_profile_summary.py $tag $O/stats $O/bench_under_rocprof.json $O/fetch $O/write $O/sqa $O/sqb $O/sqc > $O/summary.txt || { tail -5 $O/summary.txt; exit 1; }
[ -d $O/calib ] && python tools/make_profile_summary.py --calib $tag $O/calib >> $O/summary.txt
grep "^{" $O/bench_under_rocprof.json > profiles/${tag}_bench_under_rocprof.json
cp profiles/${tag}_*.csv profiles/${tag}_*.json profiles/current.json $O/      # (gpurun merges gpurun_out/ back, not profiles/)
rm -rf $O/stats $O/fetch $O/write $O/sqa $O/sqb $O/sqc $O/calib
head -80 $O/summary.txt
