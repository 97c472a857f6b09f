"""Instruction mix of one unrolled query tile of a kernel in a -save-temps .s (between the first dQ MFMA of tile n and of tile n+1).
    python tools/asm_tile_mix.py <file.s> <mangled kernel name> [tile] [mfma mnemonic marking a tile] [per tile]"""
import collections, sys
s = open(sys.argv[1]).read()
name = sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mark = sys.argv[4] if len(sys.argv) > 4 else 'v_mfma_f32_16x16x32'
per = int(sys.argv[5]) if len(sys.argv) > 5 else 8
i = s.index(name + ':')
j = s.index('.end_amdhsa_kernel', i)
body = s[i:j].split('\n')
m = [k for k, l in enumerate(body) if mark in l]
starts = m[::per]
seg = body[starts[n]:starts[n + 1]]
c = collections.Counter()
for l in seg:
    l = l.strip()
    if not l or l[0] in ';.' or l.endswith(':'):
        continue
    c[l.split()[0]] += 1
for k, v in c.most_common(60):
    print(v, k)
print('total', sum(c.values()))
