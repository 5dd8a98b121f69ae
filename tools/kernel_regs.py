#!/usr/bin/env python3
"""Register / spill / LDS figures of the kernels in a device assembly file (hipcc -S --cuda-device-only).
usage: python tools/kernel_regs.py file.s [name filter]"""
import re
import subprocess
import sys


def main():
    s = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    md = s[s.find('amdhsa.kernels'):]
    for k in re.split(r'\n  - ', md):
        n = re.search(r'\.name:\s+(\S+)', k)
        if not n:
            continue
        name = subprocess.run(['c++filt', n.group(1)], capture_output=True, text=True).stdout.strip()
        if flt not in name:
            continue

        def g(f):
            m = re.search(r'\.%s:\s+(\d+)' % f, k)
            return m.group(1) if m else '?'
        print('%-100s vgpr %s agpr %s spill %s sgpr %s lds %s scratch %s' % (
            name[:100], g('vgpr_count'), g('agpr_count'), g('vgpr_spill_count'), g('sgpr_count'),
            g('group_segment_fixed_size'), g('private_segment_fixed_size')))


if __name__ == '__main__':
    main()
