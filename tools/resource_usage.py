"""make -C lumillyrender_amd/csrc resource-usage 2>&1 | python3 tools/resource_usage.py > profiles/<round>_resource_usage.txt
Registers, scratch, occupancy and LDS of every kernel as hipcc reports them (checkable without a compile)."""
import re, subprocess, sys
out, cur = [], None
for l in sys.stdin:
    m = re.search(r'remark: (.*) \[-Rpass', l)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        name = t.split(':', 1)[1].strip()
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip() or name
        cur = {'name': dem.split('(')[0].replace('void ', '').replace('lr::', '')}
        out.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1)
        cur[k.strip()] = v.strip()
print('# make -C lumillyrender_amd/csrc resource-usage  (hipcc -Rpass-analysis=kernel-resource-usage, gfx950, ROCm 7.2)')
print('%-44s %5s %5s %8s %4s %7s %7s %7s' % ('kernel', 'VGPR', 'SGPR', 'scratchB', 'occ', 'sgprSp', 'vgprSp', 'LDS B'))
for c in out:
    print('%-44s %5s %5s %8s %4s %7s %7s %7s' % (c['name'][:44], c.get('VGPRs', '?'), c.get('TotalSGPRs', '?'), c.get('ScratchSize [bytes/lane]', '?'),
                                                  c.get('Occupancy [waves/SIMD]', '?'), c.get('SGPRs Spill', '?'), c.get('VGPRs Spill', '?'), c.get('LDS Size [bytes/block]', '?')))
