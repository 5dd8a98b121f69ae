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
