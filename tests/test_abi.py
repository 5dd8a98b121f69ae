"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every symbol that
include/dlwp_hip.h declares; argument validation that needs no device."""
import ctypes

import pytest

from dlwp_amd import _lib


def test_library_exports_every_declared_symbol():
    names = _lib.declared_symbols()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(_lib.lib, n)]
    assert not missing, missing


def test_every_declared_symbol_is_bound_with_a_signature():
    for n in _lib.declared_symbols():
        fn = getattr(_lib.lib, n)
        assert fn.argtypes is not None, '%s has no ctypes signature in dlwp_amd/_lib.py' % n


def test_version_and_error_string():
    assert _lib.lib.dlwp_version() >= 100
    assert isinstance(_lib.lib.dlwp_last_error(), bytes)


def test_struct_layouts_match_the_header():
    # sizes implied by include/dlwp_hip.h (all-int structs, no padding)
    assert ctypes.sizeof(_lib.Shape4) == 16
    assert ctypes.sizeof(_lib.Pad2d) == 24
    assert ctypes.sizeof(_lib.Conv2d) == 5 * 4 + 24 + 10 * 4     # ... src_mode, out_pool, out_d2s, lstm_f, lstm_rec_act
    # + aux[4], then the second source of a whole-step op: src2, w2, xs2_c, conv2
    assert ctypes.sizeof(_lib.Op) == 5 * 4 + 16 + ctypes.sizeof(_lib.Conv2d) + 24 + 4 * 4 + 3 * 4 + ctypes.sizeof(_lib.Conv2d)


def test_conv_out_shape_and_validation_without_a_device():
    from dlwp_amd import ops
    cd = ops.make_conv(32, 3, 3, dil=2, halo=ops.make_pad(2, 2, 2, 2, ops.PAD_ZERO, ops.PAD_WRAP), act=ops.ACT_TANH)
    ys = ops.conv_out_shape(_lib.Shape4(3, 4, 88, 180), cd)
    assert (ys.n, ys.c, ys.h, ys.w) == (3, 32, 88, 180)
    cd.src_mode = ops.SRC_MAXPOOL2
    ys = ops.conv_out_shape(_lib.Shape4(3, 4, 88, 180), cd)
    assert (ys.h, ys.w) == (44, 90)
    cd.src_mode = ops.SRC_UPSAMPLE2
    ys = ops.conv_out_shape(_lib.Shape4(3, 4, 22, 45), cd)
    assert (ys.h, ys.w) == (44, 90)
    # the reference's periodic slices do not tile: halo > axis is an error (custom.py:197-200)
    bad = ops.make_conv(4, 3, 3, halo=ops.make_pad(0, 0, 7, 7, ops.PAD_ZERO, ops.PAD_WRAP))
    with pytest.raises(_lib.DlwpError, match='column halo'):
        ops.conv_out_shape(_lib.Shape4(1, 1, 5, 6), bad)
    with pytest.raises(_lib.DlwpError, match='larger than the padded input'):
        ops.conv_out_shape(_lib.Shape4(1, 1, 3, 3), ops.make_conv(4, 5, 5))
    with pytest.raises(_lib.DlwpError, match='output channel window'):
        ops.conv_out_shape(_lib.Shape4(1, 1, 8, 8), ops.make_conv(4, 3, 3, out_c_off=2, out_c_total=4))
    # interleaved phase stores: 4 F channels -> F fields at twice the resolution, window counted in fields
    ys = ops.conv_out_shape(_lib.Shape4(2, 8, 10, 12), ops.make_conv(16, 3, 3, halo=ops.make_pad(1, 1, 1, 1), out_c_off=1,
                                                                   out_c_total=5, out_d2s=True))
    assert (ys.n, ys.c, ys.h, ys.w) == (2, 4, 20, 24)
    with pytest.raises(_lib.DlwpError, match='out_d2s'):
        ops.conv_out_shape(_lib.Shape4(1, 1, 8, 8), ops.make_conv(6, 3, 3, out_d2s=True))


def test_compiled_tile_configurations_cover_the_unet_layers():
    from dlwp_amd import ops
    cfgs = ops.conv_configs()
    kinds = {(c[0], c[1]) for c in cfgs}
    assert {(3, 1), (3, 2), (5, 1)} <= kinds
    for c in cfgs:
        ks, dil, th, tw, waves, fa, bnf, ck, pool, lds, flags = c
        pixels = th * tw if bnf > 0 else th * tw // (-bnf)          # packed-N instances tile super-pixels
        if fa == 0:                                                 # Winograd instance: 2x2 tiles, one fragment / wave
            pixels, fa = th * tw // 4, 1
        assert pixels <= 16 * fa * waves and lds <= 160 * 1024 and ck % 4 == 0


def test_bench_kernel_symbols_match_the_committed_profiles():
    """bench.py quotes rocprofv3's average duration and the PMC traffic of the dominant kernel from profiles/ by kernel
    symbol.  The symbol it composes from a tile configuration has to be the one the library really emits (template
    arguments included), or the lookup silently falls back to an older summary: every forward kernel of the newest
    same-source kernel-stats summary must be found by the name bench.py would compose for it."""
    import csv, glob, json, os, re, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import bench
    sha = bench.kernel_source_hash()
    found = 0
    # (the summary of the BENCH command: r<N>_kernel_stats -- since r6 the training / config-4 / estimator summaries carry meta files too)
    for meta_file in sorted((f for f in glob.glob(os.path.join(ROOT, 'profiles', '*kernel_stats.meta.json'))
                            if re.match(r'^r\d+[a-z]?_kernel_stats\.meta\.json$', os.path.basename(f))), reverse=True):
        meta = json.load(open(meta_file))
        if meta.get('source_sha') != sha:
            continue
        names = [r['Name'] for r in csv.DictReader(open(meta_file[:-len('.meta.json')] + '.csv'))]
        for n in names:
            m = re.search(r'conv2d_fwd_wino_f32<WinoCfg<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), false, (true|false), false, false, '
                          r'false, (true|false), (true|false), (true|false)', n)
            if m:
                dil, th, tw, waves, bnf, ck = map(int, m.groups()[:6])
                sym = bench.config_symbol((3, dil, th, tw, waves, 0, bnf, ck, 0, 0, 0), ups=m.group(7) == 'true',
                                          x_loader=(1 if m.group(8) == 'true' else (2 if m.group(9) == 'true' else 0)) +
                                          (4 if m.group(10) == 'true' else 0))
                assert sym in n, (sym, n)
                found += 1
            m = re.search(r'conv2d_fwd_wino2_f32<WinoSplitCfg<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), false', n)
            if m:
                dil, th, tw, waves, bnf, ck = map(int, m.groups())
                sym = bench.config_symbol((3, dil, th, tw, waves, 0, bnf, ck, 0, 0, 1))
                assert sym in n, (sym, n)
                found += 1
            m = re.search(r'conv2d_fwd_mfma_f32<ConvCfg<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (true|false)', n)
            if m:
                v = list(map(int, m.groups()[:8]))
                sym = bench.config_symbol(tuple(v) + (1 if m.group(9) == 'true' else 0,))
                assert sym in n, (sym, n)
                found += 1
            m = re.search(r'conv2d_fwd_wino2s_f32<(\d+), ', n)
            if m:         # the streaming position-split kernel (launch-info config -3)
                assert "'conv2d_fwd_wino2s_f32<%d, ' % ((op.xs[0] + 7) // 8)" in open(os.path.join(ROOT, 'bench.py')).read()
                found += 1
            m = re.search(r'conv2d_fwd_few_f32<(\d+), ', n)
            if m:         # the streaming layer-1 kernel: bench.py's time_layers composes this prefix for launch-info config -2
                assert "'conv2d_fwd_few_f32<%d, ' % dil_run[0]" in open(os.path.join(ROOT, 'bench.py')).read()
                found += 1
        break
    else:
        pytest.skip('no kernel-stats summary of the current kernel source in profiles/')
    assert found >= 3
    dominant = bench.config_symbol((3, 1, 8, 32, 4, 0, 2, 8, 0, 0, 0), x_loader=5)      # column pairs + edge pairs: layers 2 and 5
    ent = bench.rocprof_launch_ms(dominant, 256)
    assert ent and ent['same_source']
    traffic, src = bench.measured_traffic(dominant, 256)
    assert traffic and 2.0e8 < traffic < 4.0e8, (traffic, src)


def test_kernel_stats_listed_for_the_current_source_carry_its_hash():
    """profiles/README.md lists, per round, the summaries 'taken on the FINAL kernel source <hash>'.  For the
    section of the CURRENT source every *_kernel_stats.csv named there must exist with a .meta.json whose source_sha is that hash --
    a CSV from an earlier state of the kernels cannot sit under the heading unnoticed."""
    import glob, json, os, re
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sha = _lib.kernel_source_hash()
    text = open(os.path.join(ROOT, 'profiles', 'README.md')).read()
    sections = re.split(r'^## ', text, flags=re.M)
    mine = [sec for sec in sections if sha in sec.split('\n', 1)[0]]
    if not mine:
        pytest.skip('profiles/README.md has no section for kernel source %s (profiles are re-taken at the end of a round)' % sha)
    names = set(re.findall(r'`(r\d+[a-z]?_[A-Za-z0-9_]*kernel_stats\.csv)`', mine[0]))
    assert names, 'the section of the current source lists no kernel-stats summary'
    for n in sorted(names):
        path = os.path.join(ROOT, 'profiles', n)
        assert os.path.exists(path), n
        meta = path[:-4] + '.meta.json'
        assert os.path.exists(meta), 'no %s' % os.path.basename(meta)
        assert json.load(open(meta)).get('source_sha') == sha, n


def test_streaming_kernel_item_order_visits_every_tile_of_every_sample_once():
    """csrc/conv_fwd_few.hip hands (tile position, sample) items to its persistent workgroups by integer arithmetic alone: sample
    groups as long as a share, positions inside a group, samples inside a position; block b -> logical index (XCD b % 8, slot);
    share = [T L / grid, T (L + 1) / grid); a ragged last group.  The same arithmetic restated here must visit every item exactly
    once for any batch, tiling and grid (the GPU tests compare the kernel's results at a handful of sizes; this covers the corners:
    one member, more workgroups than items, groups longer than the batch, 2101 members)."""
    def visit(n, npos, grid):
        total = npos * n
        group = (total + grid // 2) // grid
        group = 4 if group < 4 else (n if group > n else group)
        seen = set()
        for b in range(grid):
            xcd, idx, q, r = b & 7, b >> 3, grid >> 3, grid & 7
            lidx = (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + idx
            it = total * lidx // grid
            left = total * (lidx + 1) // grid - it
            if left <= 0:
                continue
            per = npos * group
            sg, rem = divmod(it, per)
            g0 = sg * group
            gs = min(group, n - g0)
            pos, sn = divmod(rem, gs)
            for _ in range(left):
                assert 0 <= pos < npos and 0 <= g0 + sn < n and gs > 0
                assert (pos, g0 + sn) not in seen
                seen.add((pos, g0 + sn))
                sn += 1
                if sn == gs:
                    sn, pos = 0, pos + 1
                    if pos == npos:
                        pos, g0 = 0, g0 + gs
                        gs = min(group, n - g0)
        assert len(seen) == total, (n, npos, grid)
    for n in list(range(1, 20)) + [63, 64, 65, 200, 256, 257, 1031, 2101]:
        for npos in (1, 6, 66):
            for grid in (1, 7, 30, 768, 1024):
                visit(n, npos, min(grid, npos * n))


def test_every_stream_taking_export_is_taped_or_marks_the_tape_foreign():
    """The training step's tape (csrc/tape.hip) replays the entry points that carry DLWP_TAPE; any OTHER launch a recording thread
    issues would silently be missing from the replay.  Structural guard: every exported function that takes a stream
    starts with DLWP_TAPE* (recorded) or DLWP_UNTAPED (marks the tape foreign: dlwp_train_step_create then refuses the step).  A new
    entry point has to pick one."""
    import glob, os, re
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    declared = set(_lib.declared_symbols())
    allowed = {'dlwp_train_step_launch',            # refuses outright while the thread records (tape.hip)
               'dlwp_copy_many'}                    # records itself by hand (dlwp_tape_push: its pointer tables are copied)
    seen, bare = set(), []
    for f in sorted(glob.glob(os.path.join(ROOT, 'dlwp_amd', 'csrc', '*.hip'))):
        s = open(f).read()
        for m in re.finditer(r'^(?:extern "C" )?int (dlwp_\w+)\(([^)]*)\)\s*\{', s, re.M | re.S):
            name, args = m.group(1), m.group(2)
            if name not in declared or 'void* stream' not in args:
                continue
            seen.add(name)
            body = s[m.end():s.index('\n}\n', m.end())]
            if 'DLWP_TAPE' not in body and 'DLWP_UNTAPED(%s)' % name not in body and name not in allowed:
                bare.append(name)
    assert len(seen) >= 50, len(seen)
    assert not bare, bare


def test_crash_message_reaches_stdout_when_the_process_aborts():
    """dlwp_set_crash_message: bench.py parks its line there before the first one-shot exchange between real GPUs (a GPU memory fault
    makes the HSA runtime abort() the process); a parked text is written to stdout by the signal handler, a cleared one is not."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys\nsys.path.insert(0, %r)\nfrom dlwp_amd import _lib\n"
            "_lib.lib.dlwp_set_crash_message(b'{\"parked\": 1}')\n%sos.abort()\n")
    p = subprocess.run([sys.executable, '-c', code % (root, '')], capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and p.stdout.strip().endswith('{"parked": 1}')
    p = subprocess.run([sys.executable, '-c', code % (root, '_lib.lib.dlwp_set_crash_message(None)\n')], capture_output=True, text=True,
                       timeout=120)
    assert p.returncode != 0 and 'parked' not in p.stdout


def test_finalisers_bury_and_safe_points_drain():
    """dlwp_amd/_lib.py: objects that own hipGraphs are never destroyed from a finaliser -- __del__ buries the handle, the next safe
    point drains the graveyard.  (Host logic: a null handle is what dlwp_*_destroy accepts without a GPU.)"""
    import ctypes
    from dlwp_amd import _lib
    from dlwp_amd.training import _StepHandle
    from dlwp_amd.engine import RolloutGraph
    _lib.drain_graveyard()
    assert _lib._graveyard == []
    _lib.bury('step', None)                                   # nothing to bury
    assert _lib._graveyard == []
    s, g = _StepHandle(ctypes.c_void_p(0)), RolloutGraph(ctypes.c_void_p(0), None, None)
    del s, g                                                  # the finalisers run here: nothing is destroyed, two handles wait
    assert sorted(k for k, _ in _lib._graveyard) == ['rollout', 'step']
    _lib.drain_graveyard()
    assert _lib._graveyard == []
    g = RolloutGraph(ctypes.c_void_p(0), None, None)
    g.close()                                                 # an explicit close destroys at once and leaves nothing behind
    del g
    assert _lib._graveyard == []


def test_bench_counts_the_loader_variants_of_one_tile_configuration_as_one_kernel():
    """bench.py's roofline groups the input-loader variants of a Winograd tile configuration (same loop, same bits): the family
    name is a prefix of every variant's symbol, other kernels are their own family."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    cfg = (3, 1, 8, 32, 4, 0, 2, 8, 0, 0, 0)
    syms = [bench.config_symbol(cfg, x_loader=k) for k in (0, 1)]
    fam = {bench.kernel_family(s) for s in syms}
    assert len(fam) == 1 and all(s.startswith(next(iter(fam))) for s in syms) and syms[0] != syms[1]
    ups = bench.config_symbol((3, 1, 8, 32, 4, 0, 4, 8, 0, 0, 0), ups=True, x_loader=2)
    assert bench.kernel_family(ups) not in fam and ups.startswith(bench.kernel_family(ups))
    assert bench.kernel_family('conv2d_fwd_few_f32<2, ') == 'conv2d_fwd_few_f32<2, '
