"""Static instruction mix of one kernel from the -S output of hipcc (a rough guide to where the VALU work is).
    hipcc ... --cuda-device-only -S -o /tmp/zr_camera.s csrc/zr_camera.hip;  python tools/isa_mix.py /tmp/zr_camera.s <mangled-prefix> [start-line end-line]"""
import collections, sys
lines = open(sys.argv[1]).read().split('\n')
prefix = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and ':' in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
if len(sys.argv) > 4:
    start, end = int(sys.argv[3]), int(sys.argv[4])
ops = collections.Counter()
for line in lines[start + 1:end]:
    line = line.strip()
    if not line or line[0] in ';.' or line.split()[0].endswith(':'):
        continue
    ops[line.split()[0]] += 1
print(prefix, 'lines', start, end, 'total', sum(ops.values()), 'valu', sum(v for k, v in ops.items() if k.startswith('v_')),
      'salu', sum(v for k, v in ops.items() if k.startswith('s_')), 'mem', sum(v for k, v in ops.items() if k.split('_')[0] in ('global', 'flat', 'ds', 'scratch', 'buffer')))
for k, v in ops.most_common(int(sys.argv[5]) if len(sys.argv) > 5 else 40):
    print('  ', k, v)
