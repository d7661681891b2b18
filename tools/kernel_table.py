"""Per-kernel resource table from hipcc -S output: VGPRs, SGPRs, scratch bytes, LDS bytes, static instruction counts.

    python tools/kernel_table.py [source.hip ...] [--flags "-DX=1"] [--diff other.json] [--json out.json]

Default sources: every *.hip under zeldaengine_amd/csrc.  With --diff, prints only the kernels whose numbers moved against a table
saved earlier with --json: how a refactor that must not change a kernel (a device function factored out, a file split) is checked
before any GPU time is spent on it.
"""
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-gpu-rdc", "-D__HIP_PLATFORM_AMD__",
         "--cuda-device-only", "-S"]


def table(src, extra):
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-x", "hip"] + FLAGS + extra + ["-o", f.name, src])
        text = open(f.name).read()
    out = {}
    demangle = {}
    names = re.findall(r"^\s*\.amdhsa_kernel (\S+)", text, re.M)
    if names:
        d = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
        demangle = dict(zip(names, d))
    for name in names:
        body = text[text.index("\n" + name + ":"):]
        body = body[:body.index(".Lfunc_end")]
        n = v = sa = m = 0
        for line in body.split("\n")[1:]:
            t = line.strip()
            if not t or t[0] in ";." or t.split()[0].endswith(":"):
                continue
            op = t.split()[0]
            n += 1
            v += op.startswith("v_")
            sa += op.startswith("s_")
            m += op.split("_")[0] in ("global", "flat", "ds", "scratch", "buffer")
        desc = text[text.index(".amdhsa_kernel " + name):]
        desc = desc[:desc.index(".end_amdhsa_kernel")]

        def field(k, default=0):
            mm = re.search(r"\.amdhsa_%s (\d+)" % k, desc)
            return int(mm.group(1)) if mm else default
        short = re.sub(r"\(Zr.*|\(unsigned.*|\(XkI.*|\(float.*", "", demangle.get(name, name)).replace("void ", "")
        out[short] = {"vgpr": field("next_free_vgpr"), "sgpr": field("next_free_sgpr"), "scratch": field("private_segment_fixed_size"),
                      "lds": field("group_segment_fixed_size"), "insts": n, "valu": v, "salu": sa, "mem": m,
                      "flat": len(re.findall(r"^\s*flat_", body, re.M))}
    return out


def main():
    args = sys.argv[1:]
    extra, diff, save, srcs = [], None, None, []
    while args:
        a = args.pop(0)
        if a == "--flags":
            extra = args.pop(0).split()
        elif a == "--diff":
            diff = json.load(open(args.pop(0)))
        elif a == "--json":
            save = args.pop(0)
        else:
            srcs.append(a)
    if not srcs:
        srcs = sorted(glob.glob(os.path.join(ROOT, "zeldaengine_amd", "csrc", "*.hip")))
    t = {}
    for s in srcs:
        t.update(table(s, extra))
    if save:
        json.dump(t, open(save, "w"), indent=1, sort_keys=True)
    keys = ["vgpr", "sgpr", "scratch", "lds", "insts", "valu", "salu", "mem", "flat"]
    print("%-70s %s" % ("kernel", " ".join("%7s" % k for k in keys)))
    for name in sorted(t):
        if diff is not None:
            if diff.get(name) == t[name]:
                continue
            was = diff.get(name)
            print("%-70s %s" % (name[:70], " ".join("%7d" % t[name][k] for k in keys)))
            if was:
                print("%-70s %s" % ("   was", " ".join("%7d" % was[k] for k in keys)))
        else:
            print("%-70s %s" % (name[:70], " ".join("%7d" % t[name][k] for k in keys)))
    if diff is not None:
        gone = [n for n in diff if n not in t]
        if gone:
            print("gone:", ", ".join(gone))


if __name__ == "__main__":
    main()
