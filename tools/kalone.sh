# usage (GPU box): bash tools/kalone.sh lib1.so ...   per-kernel ALONE times (one stream) of the camera pass for library variants - also ones
# that draw a wrong frame on purpose (timing probes): nothing is compared
for lib in "$@"; do
  export ZELDA_RENDER_LIB=$GRAFT_REPO_ROOT/zeldaengine_amd/$lib
  ( cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_ka && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ka -o s -- python3 $GRAFT_REPO_ROOT/bench.py --serial --steps 40 --warmup 10 --no-cpu-baseline --no-extras "${KALONE_ARGS[@]}" > /dev/null 2>&1 )
  ( cd $GRAFT_REPO_ROOT && python - "$lib" <<PY
import csv, glob, sys
f = glob.glob("gpurun_out/prof_ka/**/*kernel_stats.csv", recursive=True)[0]
rows = {}
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0].replace("void ", "")
    rows[n] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
frames = rows["k_frame_begin"][0]
print(sys.argv[1] + ": " + "  ".join("%s %.1f" % (n.replace("k_", ""), rows[n][0] * rows[n][1] / frames) for n in sorted(rows) if n.startswith(("k_geom", "k_scan_tri", "k_index", "k_tile<0", "k_resolve", "k_hiz", "k_select", "k_plan", "k_cull_box<0"))))
PY
  )
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_ka
done
