#!/bin/bash
# tools/isa.sh <file.hip> [kernel-name-substring]: gfx950 ISA of one translation unit -> /tmp/t/<file>.s, resource summary,
# and (with a substring) the compact instruction stream of the matching kernels.
set -e
F=$1; B=$(basename $F .hip); mkdir -p /tmp/t
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function --cuda-device-only -S /root/repo/spike2former_amd/csrc/$B.hip -o /tmp/t/$B.s 2>&1 | grep -E "error|warning: [^a]" || true
grep -E "^\s+\.(name|vgpr_count|vgpr_spill_count|group_segment_fixed_size):" /tmp/t/$B.s | paste - - - - | awk '{print $4, "vgpr", $6, "spill", $8, "lds", $2}' | sed 's/_ZN12_GLOBAL__N_1//' | { [ -n "$2" ] && grep "$2" || cat; }
if [ -n "$2" ]; then python3 - "$B" "$2" <<'PY'
import sys, itertools, re
s = open(f'/tmp/t/{sys.argv[1]}.s').read()
for m in re.finditer(r'^(_Z\S*' + re.escape(sys.argv[2]) + r'\S*):', s, flags=re.M):
    name = m.group(1); i = m.start(); j = s.index('.Lfunc_end', i)
    seq = []
    for l in s[i:j].split('\n')[1:]:
        t = l.strip()
        if not t or t.startswith(';'): continue
        if t.startswith('.LBB'): seq.append('\n ' + t.split()[0]); continue
        if t.startswith('.'): continue
        op = t.split()[0]
        if op == 's_waitcnt': op = t.split(';')[0].strip().replace('s_waitcnt ', 'W:')
        elif 'mfma' in op: op = 'MFMA'
        elif op.startswith('v_mov'): op = 'vmov'
        elif op.startswith('v_'): op = 'V'
        elif op.startswith('s_') and not re.match(r's_(barrier|cbranch|branch|endpgm)', op): op = 'S'
        seq.append(op)
    out = []
    for k, g in itertools.groupby(seq):
        n = len(list(g)); out.append(f"{k}x{n}" if n > 1 else k)
    print(name[:110]); print(' '.join(out)[:int(sys.argv[3]) if len(sys.argv) > 3 else 3500])
PY
fi
