# usage: bash tools/resusage.sh <unit> [extra hipcc flags]   — registers, spills, occupancy, LDS of every kernel of orbit_amd/csrc/<unit>.hip
u=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize "$@" -Rpass-analysis=kernel-resource-usage -c orbit_amd/csrc/$u.hip -o /tmp/resusage_$u.o 2>&1 | python3 -c "
import re,sys,subprocess
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r'remark: +(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (.*?) \[-Rpass', l)
    if not m: continue
    k,v=m.group(1),m.group(2)
    if k=='Function Name': cur=v; rows[cur]={}
    elif cur: rows[cur][k]=v
for n,r in rows.items():
    d=subprocess.run(['c++filt',n],capture_output=True,text=True).stdout.strip()
    d=re.sub(r'orbit::\(anonymous namespace\)::|void |\(orbit::.*','',d)
    print(f\"{d[:58]:58s} VGPR {str(r.get('VGPRs')):>4s} SGPR {str(r.get('SGPRs')):>4s} spillS {str(r.get('SGPRs Spill')):>3s} spillV {str(r.get('VGPRs Spill')):>3s} scratch {str(r.get('ScratchSize [bytes/lane]')):>4s} occ {str(r.get('Occupancy [waves/SIMD]')):>2s} LDS {str(r.get('LDS Size [bytes/block]'))}\")
"
