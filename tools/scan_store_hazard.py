#!/usr/bin/env python3
"""Scan device assembly (hipcc -S --cuda-device-only) for the store-data hazard met in r3: a vector-ALU
instruction that WRITES a data register of a 12- / 16-byte buffer / global / flat / scratch store within the next few issue
slots.  hipcc pads one wait state behind such stores only when they carry no SGPR soffset; on gfx950, under load, the stored data
came out overwritten with 0 wait states behind a `buffer_store_dwordx4 ... sN offen`.
usage: python tools/scan_store_hazard.py file.s [...]   (prints kernel, store, writer, wait states in between)"""
import re
import subprocess
import sys

STORE = re.compile(r'^\s*(buffer_store_dwordx[34]|global_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34])\s+(.*)$')
VREG = re.compile(r'v\[(\d+):(\d+)\]|v(\d+)')


def regs(tok):
    m = VREG.fullmatch(tok.strip().rstrip(','))
    if not m:
        return set()
    if m.group(1):
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return {int(m.group(3))}


def main():
    bad = 0
    for path in sys.argv[1:]:
        kernel = '?'
        lines = open(path).read().split('\n')
        for i, l in enumerate(lines):
            if l and not l.startswith(('\t', ' ', '.', ';')) and l.rstrip().endswith(':') is False and ':' in l and l[0] == '_':
                kernel = l.split(':')[0]
            m = STORE.match(l)
            if not m:
                continue
            ops = [t.strip() for t in m.group(2).split(',')]
            # data operand: buffer_store: first; global_store: second (addr, data, saddr); flat: second; scratch: second-ish
            data = regs(ops[0]) if m.group(1).startswith('buffer') else (regs(ops[1]) if len(ops) > 1 else set())
            if len(data) < 3:
                continue
            waits = 0
            for j in range(i + 1, min(i + 8, len(lines))):
                t = lines[j].strip()
                if not t or t.startswith((';', '.')) or t.endswith(':'):
                    continue
                if t.startswith('s_nop'):
                    waits += int(t.split()[1]) + 1
                    continue
                op = t.split()[0]
                if op.startswith('v_') and not op.startswith(('v_cmp', 'v_cmpx', 'v_nop', 'v_mfma', 'v_readlane', 'v_readfirstlane')):
                    dst = regs(t.split(None, 1)[1].split(',')[0]) if len(t.split(None, 1)) > 1 else set()
                    if dst & data:
                        name = subprocess.run(['c++filt', kernel], capture_output=True, text=True).stdout.strip()[:90]
                        print('%s | %s | %s | %d wait states' % (name, l.strip()[:60], t[:50], waits))
                        bad += 1
                        break
                if op.startswith(('s_waitcnt', 's_barrier', 's_cbranch', 's_branch', 's_endpgm')):
                    break
                waits += 1
                if waits >= 5:
                    break
    print('%d store(s) with a vector write to their data registers within 5 issue slots' % bad)


if __name__ == '__main__':
    main()
