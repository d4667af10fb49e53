#!/bin/bash
# tools/resource_usage.sh <file.hip> [extra flags]: VGPR / SGPR / spill / occupancy summary of every kernel of a translation unit (CPU only)
f=$1; shift
extra=""
case "$(basename $f)" in bgk_fused2*.hip) extra="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-comment $extra "$@" \
  -Rpass-analysis=kernel-resource-usage -c -o /dev/null $f 2>&1 | python3 -c "
import sys,re,subprocess
cur=None; rows=[]
for ln in sys.stdin:
    m=re.search(r'remark: Function Name: (\S+)',ln)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    m=re.search(r'remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): (\d+)',ln)
    if m and cur is not None: cur[m.group(1).split(' [')[0].replace(' ','')]=m.group(2)
for r in rows:
    nm=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip().replace('(anonymous namespace)::','')
    print(nm[:70].ljust(72),' '.join(f'{k}={v}' for k,v in r.items() if k!='name'))
"
