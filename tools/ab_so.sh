#!/bin/bash
# Same-box A/B of two builds of the library: orcai_amd/liborcai_hip.so (variant) against orcai_amd/liborcai_hip_base.so (baseline), per-call tables of the training steps.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_so; rm -rf $O; mkdir -p $O
cd $R
for round in 1 2; do
  python3 tools/step_calls.py f32 > $O/var_f32_$round.txt 2>&1; python3 tools/step_calls.py f16 set3 > $O/var_f16_$round.txt 2>&1
  cp orcai_amd/liborcai_hip.so /tmp/variant.so; cp orcai_amd/liborcai_hip_base.so orcai_amd/liborcai_hip.so
  python3 tools/step_calls.py f32 > $O/base_f32_$round.txt 2>&1; python3 tools/step_calls.py f16 set3 > $O/base_f16_$round.txt 2>&1
  cp /tmp/variant.so orcai_amd/liborcai_hip.so
done
python3 - <<PY
import re,collections
def load(f):
    d=collections.OrderedDict()
    for l in open(f):
        m=re.match(r'(\S+)\s+(.{58})\s+([0-9.]+)',l)
        if m and not l.startswith('launcher'):
            k=(m.group(1),m.group(2).strip(),l[110:].strip()); d[k]=float(m.group(3))
    return d
for prec in ('f32','f16'):
    v=[load(f"$O/var_{prec}_{r}.txt") for r in (1,2)]; b=[load(f"$O/base_{prec}_{r}.txt") for r in (1,2)]
    tv=tb=0
    print(prec)
    for k in v[0]:
        if k in b[0] and k in v[1] and k in b[1]:
            mv=min(v[0][k],v[1][k]); mb=min(b[0][k],b[1][k]); tv+=mv; tb+=mb
            if abs(mv-mb)>0.004: print(f"  {k[0][:28]:28s} {k[1][:52]:52s} base {mb:.3f} variant {mv:.3f}  {k[2][:40]}")
    print(f"  sum base {tb:.3f} variant {tv:.3f}")
PY
