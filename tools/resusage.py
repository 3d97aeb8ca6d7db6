"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr log) per kernel."""
import re
import sys

txt = open(sys.argv[1]).read()
for b in txt.split('Function Name: ')[1:]:
    name = b.split()[0]
    def g(k):
        m = re.search(re.escape(k) + r': (\d+)', b)
        return int(m.group(1)) if m else -1
    m = re.search(r'sweep_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E', name)
    short = "sweep<G=%s,R=%s,W=%s,L=%s>" % m.groups() if m else name[:40]
    print("%-34s vgpr=%d agpr=%d sgpr=%d scratch=%d occ=%d vspill=%d lds=%d" % (
        short, g('VGPRs'), g('AGPRs'), g('SGPRs'), g('ScratchSize [bytes/lane]'), g('Occupancy [waves/SIMD]'),
        g('VGPRs Spill'), g('LDS Size [bytes/block]')))
