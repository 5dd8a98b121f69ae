#!/usr/bin/env python3
"""Memory waits right behind the loads they wait for, per kernel of a built object (r6).

The lesson of profiles/r6_wgrad_cb_knockout.txt: a select (or any ALU use) written directly behind a prefetch load makes the compiler
put `s_waitcnt vmcnt(0)` a few instructions after the load -- a full memory latency exposed on every load of a software-pipelined
loop, and a wait for every load issued before it.  Knock-out builds see it only as "the loads cost N cycles to issue".  It also counts WATERFALL loops: a buffer instruction whose scalar offset lives in a VGPR (a value that differs inside the wave, or
one the compiler cannot prove uniform) is wrapped in readfirstlane / compare / execute-under-mask / repeat.  This
scanner lists, for every kernel, the `s_waitcnt vmcnt(n)` that force the completion of a vector memory load issued at most --near
instructions earlier (the counter decrements in issue order: `vmcnt(n)` completes everything but the newest n; a wait for the OLDEST of
many loads in flight right after issuing new ones is a working pipeline and is not listed).

    python tools/isa_waits.py dlwp_amd/csrc/build/conv_bwd.o [--kernel wino_cb] [--near 6] [--all]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'


def disassemble(obj):
    tmp = tempfile.mkdtemp(prefix='isa_waits_')
    fat, co = os.path.join(tmp, 'fat'), os.path.join(tmp, 'co')
    subprocess.check_call(['objcopy', '-O', 'binary', '--only-section=.hip_fatbin', obj, fat])
    subprocess.check_call([LLVM + '/clang-offload-bundler', '--type=o', '--unbundle', '--input=' + fat,
                           '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co])
    global NOTES
    NOTES = subprocess.check_output([LLVM + '/llvm-readelf', '--notes', co], text=True)
    return subprocess.check_output([LLVM + '/llvm-objdump', '-d', '--demangle', co], text=True)


NOTES = ''


def registers():
    """{demangled kernel name: 'vgpr N (spilled M) sgpr K'} from the code object's metadata"""
    out = {}
    for k in re.split(r'\n\s+- \.agpr_count', NOTES)[1:]:
        n = re.search(r'\.name:\s+(\S+)', k)
        if not n:
            continue
        name = subprocess.run(['c++filt', n.group(1)], capture_output=True, text=True).stdout.strip()

        def g(f):
            m = re.search(r'\.%s:\s+(\d+)' % f, k)
            return m.group(1) if m else '?'
        out[name.replace(' ', '')] = 'vgpr %s (spilled %s) sgpr %s (spilled %s)' % (g('vgpr_count'), g('vgpr_spill_count'), g('sgpr_count'), g('sgpr_spill_count'))
    return out


def kernels(txt):
    heads = [(m.start(), m.group(1)) for m in re.finditer(r'^[0-9a-f]+ <(.+)>:$', txt, re.M)]
    for k, (pos, name) in enumerate(heads):
        end = heads[k + 1][0] if k + 1 < len(heads) else len(txt)
        yield name, [x.split('//')[0].strip() for x in txt[pos:end].splitlines()[1:] if x.strip()]


def is_vload(l):
    return re.match(r'(buffer_load|global_load|flat_load|scratch_load)', l) is not None


def _scan(lines, near, is_issue, counter):
    """-> [(index, wait, distance to the newest load the wait forces to complete, loads in flight, next instruction)] for the waits
    that force a load issued at most `near` instructions earlier.  The counter decrements in issue order: `cnt(n)` completes
    everything but the newest n."""
    found, flying = [], []
    for i, l in enumerate(lines):
        if is_issue(l):
            flying.append(i)
        elif l.startswith('s_waitcnt') and counter in l:
            n = int(re.search(counter + r'\((\d+)\)', l).group(1))
            if len(flying) > n:
                forced = flying[:len(flying) - n]
                if i - forced[-1] <= near:
                    found.append((i, l, i - forced[-1], len(flying), lines[i + 1] if i + 1 < len(lines) else ''))
                flying = flying[len(flying) - n:]
    return found


def scan_lds(lines, near):
    """LDS reads (scalar loads share lgkmcnt; they are rare inside the loops this is for)"""
    return _scan(lines, near, lambda l: l.startswith('ds_read'), 'lgkmcnt')


def scan(lines, near):
    return _scan(lines, near, is_vload, 'vmcnt')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('obj')
    ap.add_argument('--kernel', default='')
    ap.add_argument('--near', type=int, default=6)
    ap.add_argument('--lds', action='store_true', help='also list lgkmcnt waits right behind the ds_read they wait for')
    ap.add_argument('--all', action='store_true', help='list every hit instead of the per-kernel count and first few')
    a = ap.parse_args()
    txt = disassemble(a.obj)
    regs = registers()
    for name, lines in kernels(txt):
        if a.kernel not in name:
            continue
        mf = [i for i, l in enumerate(lines) if 'v_mfma' in l]
        hits = scan(lines, a.near)
        inside = [h for h in hits if mf and mf[0] < h[0] < mf[-1]]
        scratch = sum(1 for l in lines if l.startswith('scratch_'))
        # a memory instruction whose scalar operand the compiler could not prove wave-uniform: readfirstlane, compare, the
        # instruction under the matching lanes, repeat for the rest
        falls = sum(1 for i, l in enumerate(lines[:-1]) if l.startswith('s_xor_b64 exec') and lines[i + 1].startswith('s_cbranch_execnz'))
        print('%s\n   %s; %d instructions, %d MFMAs, %d scratch accesses, %d waterfall loops; waits right behind their load: %d (%d '
              'between the first and the last MFMA)' % (name[:150], regs.get(name.replace(' ', ''), '?'), len(lines), len(mf), scratch, falls, len(hits),
                                                       len(inside)))
        for h in (hits if a.all else inside[:6]):
            print('      +%d %-22s %d behind the load, %d loads out | then %s' % (h[0], h[1], h[2], h[3], h[4][:70]))
        if a.lds:
            lh = [h for h in scan_lds(lines, a.near) if mf and mf[0] < h[0] < mf[-1]]
            print('   LDS waits right behind their read, between the first and the last MFMA: %d' % len(lh))
            for h in (lh if a.all else lh[:6]):
                print('      +%d %-22s %d behind the read, %d reads out | then %s' % (h[0], h[1], h[2], h[3], h[4][:70]))


if __name__ == '__main__':
    sys.exit(main())
