#!/bin/bash
# Instruction-stream skeleton of one kernel of a csrc/*.hip file:  bash tools/isa.sh <file.hip> <kernel-name-regex> [full]
# (barriers, waits, memory instructions, MFMAs and LDS reads counted in runs; `full` prints the whole stream)
cd "$(dirname "$0")/../spike2former_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics --cuda-device-only -S "$1" -o /tmp/isa_$$.s 2>/dev/null
L=$(grep -nE "^_Z.*($2).*:" /tmp/isa_$$.s | head -1 | cut -d: -f1)
[ -z "$L" ] && { echo "no kernel matches $2"; grep -nE "^_Z.*:" /tmp/isa_$$.s | cut -c1-160; exit 1; }
tail -n +"$L" /tmp/isa_$$.s | awk '/s_endpgm/{print; exit} {print}' > /tmp/isa_$$.k
if [ "$3" = full ]; then cat /tmp/isa_$$.k; else
grep -nE "s_barrier|s_waitcnt|v_mfma|ds_read|ds_write|global_load|buffer_load|global_store|s_setprio|^\.LBB|global_atomic|s_endpgm" /tmp/isa_$$.k |
  awk '{ if ($0 ~ /v_mfma/) {m++; next} if ($0 ~ /ds_read/) {r++; next} if (m>0) {print "   ... " m " mfma"; m=0} if (r>0) {print "   ... " r " ds_read"; r=0} print }'
fi
rm -f /tmp/isa_$$.s /tmp/isa_$$.k
