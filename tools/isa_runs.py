#!/usr/bin/env python3
"""isa_runs.py -- how the matrix instructions and the vector work alternate inside a kernel's loops (on gfx950 every switch
between an fp32 MFMA and a vector instruction costs ~11 cycles, so the count of RUNS matters as much as the instruction count).
Usage: isa_runs.py <library.so | object> <kernel-name substring> [--all]
Disassembles the gfx950 code object (llvm-objdump), finds the backward branches of the kernel, and prints for the body of every loop
the sequence  M<n> V<n> L<n> G<n> S<n>  (matrix / vector ALU / LDS / global-buffer / scalar+other instructions in issue order) plus totals."""
import re
import subprocess
import sys

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'


def kind(op):
    if op.startswith('v_mfma') or op.startswith('v_smfmac'):
        return 'M'
    if op.startswith('ds_'):
        return 'L'
    if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')):
        return 'G'
    if op.startswith('v_'):
        return 'V'
    return 'S'


def main():
    lib, pat = sys.argv[1], sys.argv[2]
    # the .so carries an offload bundle: extract the gfx950 code object first
    import tempfile, os
    tmp = tempfile.mkdtemp()
    co = os.path.join(tmp, 'co')
    r = subprocess.run(['/opt/rocm/lib/llvm/bin/clang-offload-bundler', '--type=o', '--unbundle', '--input=' + lib,
                        '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co], capture_output=True, text=True)
    src = co if r.returncode == 0 and os.path.getsize(co) else lib
    if src == lib:
        # a host .so: the bundle sits in section .hip_fatbin
        fat = os.path.join(tmp, 'fat')
        subprocess.run(['objcopy', '-O', 'binary', '--only-section=.hip_fatbin', lib, fat], check=True)
        subprocess.run(['/opt/rocm/lib/llvm/bin/clang-offload-bundler', '--type=o', '--unbundle', '--input=' + fat,
                        '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co], check=True)
        src = co
    txt = subprocess.run([OBJDUMP, '-d', '--demangle', src], capture_output=True, text=True).stdout
    blocks = re.split(r'\n(?=[0-9a-f]+ <)', txt)
    for blk in blocks:
        head = blk.split('\n', 1)[0]
        if pat not in head or '>:' not in head:
            continue
        ins = []
        for line in blk.split('\n')[1:]:
            m = re.match(r'\s+(\S+)\s.*//\s*([0-9A-Fa-f]+):', line)
            if not m:
                continue
            ins.append((int(m.group(2), 16), m.group(1), line))
        addr_ix = {a: i for i, (a, _, _) in enumerate(ins)}
        print('==', head[:200], '|', len(ins), 'instructions,', sum(1 for _, o, _ in ins if kind(o) == 'M'), 'MFMA')
        loops = []
        for i, (a, op, line) in enumerate(ins):
            if op.startswith('s_cbranch') or op == 's_branch':
                m = re.search(r'<[^>]*\+0x([0-9a-f]+)>', line) or re.search(r'<[^>+]*>', line)
                tgt = None
                m2 = re.search(r'\+0x([0-9a-f]+)>', line)
                if m2:
                    base = ins[0][0]
                    tgt = base + int(m2.group(1), 16)
                if tgt is not None and tgt in addr_ix and addr_ix[tgt] < i:
                    loops.append((addr_ix[tgt], i))
        for lo, hi in loops:
            body = ins[lo:hi + 1]
            nm = sum(1 for _, o, _ in body if kind(o) == 'M')
            if nm == 0 and '--all' not in sys.argv:
                continue
            seq, runs = [], {'M': 0, 'V': 0}
            prev, n = None, 0
            for _, o, _ in body:
                k = kind(o)
                if o in ('s_nop', 's_waitcnt', 's_barrier') or o.startswith('s_'):
                    k = 'S'
                if k == prev:
                    n += 1
                else:
                    if prev:
                        seq.append('%s%d' % (prev, n))
                    prev, n = k, 1
            seq.append('%s%d' % (prev, n))
            # runs of matrix instructions ignoring scalar / LDS / memory instructions between them (only V breaks a run)
            mv = [kind(o) for _, o, _ in body if kind(o) in 'MV']
            sw = sum(1 for a, b in zip(mv, mv[1:]) if a != b)
            tot = {k: sum(1 for _, o, _ in body if kind(o) == k) for k in 'MVLGS'}
            print('  loop [%d..%d] %d instr: %s | M<->V switches %d' % (lo, hi, len(body), tot, sw))
            print('   ', ' '.join(seq))


main()
