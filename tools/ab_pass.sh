# usage (GPU box): bash tools/ab_pass.sh lib1.so lib2.so ...   A/B of library variants on what the GBuffer-write pass costs:
#   per variant (a) the two-lane benchmark frame (200 frames: value, the pass's kernels beside the other lane) and
#   (b) every kernel ALONE (one stream, rocprofv3 --kernel-trace --stats): the camera pass's kernels and their sum.
for lib in "$@"; do
  export ZELDA_RENDER_LIB=$GRAFT_REPO_ROOT/zeldaengine_amd/$lib
  ( cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > gpurun_out/abp_$lib.log 2>&1 ) || { echo "$lib FAILED"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/abp_$lib.log; exit 1; }
  ( cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_abp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_abp -o s -- python3 $GRAFT_REPO_ROOT/bench.py --serial --steps 40 --warmup 10 --no-cpu-baseline --no-extras > /dev/null 2>&1 )
  ( cd $GRAFT_REPO_ROOT && python - "$lib" <<PY
import csv, glob, json, sys
lib = sys.argv[1]
l = [x for x in open("gpurun_out/abp_%s.log" % lib) if x.startswith("{")][-1]
d = json.loads(l)
p = d["passes_ms"]
print("%-24s %8.1f Mpx/s  %.4f ms | beside: gbuffer %.1f hiz %.1f gbuffer2 %.1f resolve %.1f = pass %.1f us (frac %.4f)  lighting %.1f shadow %.1f" % (
    lib, d["value"], d["ms_per_step"], p["gbuffer"] * 1e3, p["hiz"] * 1e3, p["gbuffer2"] * 1e3, p["resolve"] * 1e3,
    d["roofline"]["kernel_ms"] * 1e3, d["roofline"]["frac"], p["lighting"] * 1e3, p["shadow"] * 1e3))
f = glob.glob("gpurun_out/prof_abp/**/*kernel_stats.csv", recursive=True)[0]
rows = {}
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0].replace("void ", "")
    rows[n] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
frames = rows["k_frame_begin"][0]
tot = 0.0
out = []
for n in sorted(rows):
    if n.startswith(("k_geom", "k_scan_tri", "k_index", "k_tile<0", "k_resolve", "k_hiz", "k_select")):
        per = rows[n][0] * rows[n][1] / frames
        if not n.startswith(("k_hiz", "k_select")) and not (n.startswith("k_geom") and n.endswith(", true>")): tot += per      # (count-only k_geom: first frame only)
        out.append("%s %.1f" % (n.replace("k_", ""), per))
print("    alone, us per frame: " + "  ".join(out) + "  | GBuffer-write pass alone %.1f us" % tot)
PY
  )
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_abp
done
