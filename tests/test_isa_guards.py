"""Static guards on the built kernels (no GPU: llvm-objdump on the objects `__graft_entry__.build()` leaves under
dlwp_amd/csrc/build).  DESIGN.md 5.21: the weight-gradient family lost 5-25 % of every launch to things only the disassembly shows --
`s_waitcnt vmcnt` right behind a prefetch load (a select or a copy written behind it), waterfall loops around buffer loads whose scalar
offset the compiler could not keep in an SGPR, spills inside matrix loops.  tools/isa_waits.py lists them; this test pins the kernels
that are clean, so that a source change (or a compiler that allocates differently) cannot bring them back unnoticed."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, 'dlwp_amd', 'csrc', 'build')
sys.path.insert(0, os.path.join(ROOT, 'tools'))

# (object, substring of the demangled kernel name) -- the kernels of the bench's forward and of the config-3 training step whose loops
# carry software-pipelined global loads
CLEAN = [
    ('conv_bwd.o', 'WgCbCfg<4, 32, 4, 2, 2, true>'),     # layer 4's weight gradient (up-sampled source, conv_wgrad_cbu_kernel.h)
    ('conv_bwd.o', 'WgCbCfg<4, 32, 2, 2, 1, true>'),
    ('conv_bwd.o', 'WgCbCfg<8, 16, 4, 2, 1, true>'),
    ('conv_fwd_k3d1.o', 'WinoCfg<1, 8, 32, 4, 2, 8, false, false, false, false, false, true, false, true>'),
    ('conv_fwd_k3d1.o', 'WinoCfg<1, 8, 32, 4, 2, 8, false, false, false, false, false, false, false, false>'),
    ('conv_fwd_k3d1.o', 'WinoCfg<1, 8, 32, 4, 4, 8, false, true, false, false, false, false, true, false>'),
    ('conv_fwd_few.o', 'conv2d_fwd_few_f32<2, 1, true, 0, 1>'),
    ('conv_fwd_bf16_o8.o', 'BfCfg<3, 2, 8, 32, 4, 4, 4, 32, false, true, true, true, true>'),   # config 4: the whole ConvLSTM2D step
]
# no waterfall loops and no scratch anywhere in these objects' matrix kernels (the plain channel-block instances keep run-time source modes
# whose pooled paths wait behind their loads by construction: they are checked for loops and spills only)
NO_LOOPS = ['conv_bwd.o', 'conv_pair.o', 'rowconv.o', 'conv_fwd_k3d1.o', 'conv_fwd_k3d2.o', 'conv_fwd_few.o', 'conv_fwd_wino2s.o',
            'conv_fwd_bf16.o', 'conv_fwd_bf16_o8.o']


def _scan(obj):
    import isa_waits as w
    path = os.path.join(BUILD, obj)
    if not os.path.exists(path) or not os.path.exists(w.LLVM + '/llvm-objdump'):
        pytest.skip('%s is not built here (or no llvm-objdump)' % obj)
    txt = w.disassemble(path)
    out = {}
    for name, lines in w.kernels(txt):
        mf = [i for i, l in enumerate(lines) if 'v_mfma' in l]
        falls = sum(1 for i, l in enumerate(lines[:-1]) if l.startswith('s_xor_b64 exec') and lines[i + 1].startswith('s_cbranch_execnz'))
        inside = [h for h in w.scan(lines, 6) if mf and mf[0] < h[0] < mf[-1]]
        out[name] = {'mfma': len(mf), 'waterfall': falls, 'waits_in_loop': inside,
                     # spills are tolerated around the matrix loop (the bf16 whole-step launch parks its prefetched cell state there), not
                     # inside it: a scratch reload waits on vmcnt, i.e. for every prefetch load in flight
                     'scratch_in_loop': sum(1 for i, l in enumerate(lines) if l.startswith('scratch_') and mf and mf[0] < i < mf[-1])}
    return out


_cache = {}


def _kernels(obj):
    if obj not in _cache:
        _cache[obj] = _scan(obj)
    return _cache[obj]


@pytest.mark.parametrize('obj,kernel', CLEAN)
def test_pipelined_loops_have_no_wait_right_behind_their_prefetch_loads(obj, kernel):
    ks = {n: v for n, v in _kernels(obj).items() if kernel in n}
    assert ks, 'no kernel %r in %s' % (kernel, obj)
    for n, v in ks.items():
        assert v['mfma'] > 0, n
        assert not v['waits_in_loop'], (n, v['waits_in_loop'][:3])
        assert v['waterfall'] == 0 and v['scratch_in_loop'] == 0, (n, v['waterfall'], v['scratch_in_loop'])


@pytest.mark.parametrize('obj', NO_LOOPS)
def test_matrix_kernels_have_no_waterfall_loops(obj):
    # (the two-fragment plain channel-block instances spill by design and are priced out by pick_wgrad: scratch is not asserted here)
    bad = {n[:120]: v['waterfall'] for n, v in _kernels(obj).items() if v['mfma'] > 0 and v['waterfall']}
    assert not bad, bad


def test_the_scanner_flags_a_wait_for_a_fresh_load_and_not_a_wait_for_the_oldest_of_many():
    """tools/isa_waits.py on synthetic listings: vmcnt decrements in issue order, `vmcnt(n)` completes all but the newest n loads.
    A wait that forces a load issued a few instructions earlier is the defect; a wait for the oldest of many loads in flight, placed
    right behind fresh loads, is a working software pipeline."""
    import isa_waits as w
    fresh = ['buffer_load_dwordx4 v[0:3], v9, s[0:3], 0 offen', 's_waitcnt vmcnt(0)', 'v_cndmask_b32_e32 v0, 0, v0, vcc']
    assert [h[:4] for h in w.scan(fresh, 6)] == [(1, 's_waitcnt vmcnt(0)', 1, 1)]
    pipe = ['buffer_load_dword v%d, v9, s[0:3], 0 offen' % k for k in range(4)] + ['v_add_f32_e32 v20, v21, v22'] * 30 + \
           ['buffer_load_dword v%d, v9, s[0:3], 0 offen' % k for k in range(4, 8)] + ['s_waitcnt vmcnt(4)', 'ds_write_b32 v30, v0']
    assert w.scan(pipe, 6) == []                                   # forces loads 0-3, issued 30+ instructions ago
    assert len(w.scan(pipe[:-2] + ['s_waitcnt vmcnt(3)', 'ds_write_b32 v30, v4'], 6)) == 1      # ... one of the fresh four as well
    lds = ['ds_read_b64 v[0:1], v9', 'ds_read_b64 v[2:3], v9 offset:8', 's_waitcnt lgkmcnt(1)', 'v_mfma_f32_16x16x4_f32 v[4:7], v0, v1, v[4:7]']
    assert [h[0] for h in w.scan_lds(lds, 3)] == [2]
    assert w.is_vload('global_load_dwordx2 v[0:1], v[2:3], off') and not w.is_vload('buffer_store_dword v0, v1, s[0:3], 0 offen')
