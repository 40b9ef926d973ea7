"""dev: instruction mix of kernels in an assembly listing (hipcc -S --cuda-device-only): python isa_count.py file.s name..."""
import re, sys, collections
txt = open(sys.argv[1]).read()
for name in sys.argv[2:]:
    m = re.search(r'^(_Z\w*%s\w*):[^\n]*\n(.*?)s_endpgm' % name, txt, re.S | re.M)
    if not m:
        print(name, "not found"); continue
    c = collections.Counter()
    for line in m.group(2).splitlines():
        line = line.strip()
        if not line or line[0] in ';.' or line.split()[0].endswith(':'):
            continue
        c[line.split()[0]] += 1
    groups = collections.Counter()
    for op, n in c.items():
        g = ('valu' if op.startswith('v_') else 'wait' if op.startswith('s_waitcnt') else 'barrier' if op.startswith('s_barrier') else 'salu' if op.startswith('s_')
             else 'lds' if op.startswith('ds_') else 'vmem' if op.split('_')[0] in ('buffer', 'global', 'scratch', 'flat') else op)
        groups[g] += n
    print(name, sum(c.values()), dict(groups))
    print('   ', c.most_common(30))
