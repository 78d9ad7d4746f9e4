#!/bin/bash
# kernel resource usage of every device unit as the compiler reports it (-Rpass-analysis=kernel-resource-usage): VGPRs, spills, scratch, occupancy
#   bash tools/resource_summary.sh > profiles/<tag>_kernel_resources.txt        (no GPU needed)
cd "$(dirname "$0")/../rmcv_amd/csrc"
for u in k_binary k_contours k_contours_w4 k_detect k_classify k_pnp; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Rpass-analysis=kernel-resource-usage -c $u.hip -o /tmp/rs_$u.o 2>&1 | c++filt | python3 -c "
import re, sys
unit = '$u'
cur = {}
def flush():
    if cur.get('name'):
        print('%-14s VGPRs %3s  AGPRs %2s  scratch %4s B/lane  occupancy %s waves/SIMD  SGPR spills %3s  VGPR spills %2s  static LDS %6s B  %s' % (
            unit, cur.get('VGPRs', '?'), cur.get('AGPRs', '?'), cur.get('ScratchSize [bytes/lane]', '?'), cur.get('Occupancy [waves/SIMD]', '?'),
            cur.get('SGPRs Spill', '?'), cur.get('VGPRs Spill', '?'), cur.get('LDS Size [bytes/block]', '?'), cur['name'][:110]))
for ln in sys.stdin:
    m = re.search(r'remark: +Function Name: (.*?) \[-Rpass', ln)
    if m:
        flush(); cur = {'name': m.group(1)}; continue
    m = re.search(r'remark: +([A-Za-z /\[\]]+?): (\S+) \[-Rpass', ln)
    if m:
        cur[m.group(1).strip()] = m.group(2)
flush()"
done
