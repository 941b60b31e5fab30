"""CPU-side checks of the C ABI: the library loads, exports what include/hpk.h declares, the ctypes mirrors
have the C sizes, and the host-only plan helper reproduces the as-coded ring multiplicities."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import REPO
from hicpeaks_amd import _lib


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    hdr = open(os.path.join(REPO, 'include', 'hpk.h')).read()
    declared = set(re.findall(r'\b(hpk_[a-z_]+)\s*\(', hdr))
    assert declared == set(_lib.ABI_SYMBOLS), declared ^ set(_lib.ABI_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.hpk_abi_version() == 3


def test_ctypes_struct_sizes_match_header(tmp_path):
    src = tmp_path / 'sz.c'
    src.write_text('#include "%s/include/hpk.h"\n#include <stdio.h>\nint main(){printf("%%zu %%zu %%zu %%zu\\n",'
                   'sizeof(hpk_result),sizeof(hpk_params),sizeof(hpk_band),sizeof(hpk_set));return 0;}\n' % REPO)
    exe = tmp_path / 'sz'
    subprocess.check_call(['gcc', str(src), '-o', str(exe)])
    sizes = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert sizes == [ctypes.sizeof(_lib.Result), ctypes.sizeof(_lib.Params), ctypes.sizeof(_lib.Band),
                     ctypes.sizeof(_lib.Set)]


def test_plan_known_answer_union():
    # SURVEY.md §8-A5 known-answer table (pw=[1,2,4], ww=[3,5,7]; m_1..m_8)
    p = _lib.make_params(_lib.MODE_HICCUPS, [1, 2, 4], [3, 5, 7], 10, 0.05, 5000000, 10000)
    steps, mk, mr = _lib.plan_rings(p)
    assert len(steps) == 18
    want = {(1, 3): [0, 1, 1, 0, 0, 0, 0, 0], (1, 5): [0, 1, 1, 1, 1, 0, 0, 0], (2, 5): [0, 1, 1, 1, 1, 0, 0, 0],
            (1, 6): [0, 2, 1, 1, 1, 1, 0, 0], (2, 6): [0, 2, 1, 1, 1, 1, 0, 0], (4, 7): [0, 3, 1, 1, 1, 1, 1, 0],
            (1, 8): [0, 4, 2, 2, 1, 1, 1, 1], (4, 8): [0, 4, 2, 2, 1, 1, 1, 1]}
    for s, m in zip(steps, mk):
        if s in want:
            assert m[1:9].tolist() == want[s], s
    # Reads = lower-left rings min(pw)+1 .. wi
    for (pi, wi), r in zip(steps, mr):
        assert r[1:11].tolist() == [1 if 2 <= rho <= wi else 0 for rho in range(1, 11)]


def test_plan_single_pair_is_textbook_donut():
    p = _lib.make_params(_lib.MODE_HICCUPS, [2], [5], 10, 0.05, 5000000, 10000)
    steps, mk, mr = _lib.plan_rings(p)
    assert steps == [(2, w) for w in range(5, 11)]
    for (pi, wi), m in zip(steps, mk):
        assert m[:11].tolist() == [1 if pi < rho <= wi else 0 for rho in range(11)]
    p = _lib.make_params(_lib.MODE_BHFDR, [2], [5], 20, 0.05, 2000000, 10000)
    steps, mk, mr = _lib.plan_rings(p)
    assert steps == [(2, w) for w in range(5, 21)]
    assert mk[-1][:21].tolist() == [1 if 2 < rho <= 20 else 0 for rho in range(21)]


@pytest.mark.parametrize('pw,ww,maxww', [([1, 2, 4], [3, 5, 7], 10), ([2], [5], 10), ([4], [7], 10), ([1, 2], [3, 5], 8),
                                         ([2, 4], [5, 7], 20), ([1], [3], 6)])
def test_innermost_box_is_shared_by_all_steps(pw, ww, maxww):
    """What hpk_stencil_s's shared inner box rests on (HpkDevPlan::first_rho): written as box terms c_rho = m_rho - m_(rho+1),
    every step's first term sits at radius min(pw) - no ring at or inside it ever counts, the ring just outside always does."""
    p = _lib.make_params(_lib.MODE_HICCUPS, pw, ww, maxww, 0.05, 5000000, 10000)
    steps, mk, mr = _lib.plan_rings(p)
    assert steps
    for (pi, wi), m in zip(steps, mk):
        m = m.tolist() + [0]
        terms = [(rho, m[rho] - m[rho + 1]) for rho in range(1, maxww + 1) if m[rho] - m[rho + 1] != 0]
        assert terms[0][0] == min(pw) and terms[0][1] < 0, (pi, wi, terms)
        assert sum(c for _, c in terms) == 0                    # the pixel's own value cancels (box_ky_d)


def test_plan_rejects_bad_arguments():
    with pytest.raises(_lib.HpkError):
        _lib.plan_rings(_lib.make_params(_lib.MODE_HICCUPS, [2], [5], 21, 0.05, 2000000, 10000))
    with pytest.raises(_lib.HpkError):
        _lib.plan_rings(_lib.make_params(_lib.MODE_BHFDR, [2, 1], [5, 3], 10, 0.05, 2000000, 10000))
    # a pair whose donut is wider than maxww contributes no step (callers.py:19)
    p = _lib.make_params(_lib.MODE_HICCUPS, [1, 4], [3, 12], 10, 0.05, 2000000, 10000)
    steps, _, _ = _lib.plan_rings(p)
    assert steps == [(1, w) for w in range(3, 11)]


def test_no_device_is_a_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_lib.HpkError) as e:
        _lib.Context(0)
    assert e.value.status == _lib.ERR_NO_DEVICE
    assert 'no CPU path' in str(e.value)


def test_chunk_bounds_match_numpy():
    lib = _lib.load()
    b = np.zeros(_lib.HPK_NB)
    assert lib.hpk_chunk_bounds(b.ctypes.data, _lib.HPK_NB) == 0
    want = np.array([np.power(2, ((i - 1) / 3.)) for i in range(1, _lib.HPK_NB + 1)])
    np.testing.assert_allclose(b, want, rtol=4e-16, atol=0)


def test_one_hip_runtime_whichever_of_hicpeaks_amd_and_torch_comes_first():
    """libhpk.so and PyTorch's ROCm wheel each ask for libamdhip64 their own way; two runtimes in one process lose the device for
    the second (round 5: worked around in the tests' conftest only).  _lib.load() makes both resolve to one copy."""
    import subprocess
    import sys
    code = ('import os, sys\n'
            'order = sys.argv[1]\n'
            'if order == "torch_first":\n'
            '    import torch\n'
            'from hicpeaks_amd import _lib\n'
            '_lib.load()\n'
            'import torch\n'
            'maps = set(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)\n'
            'print(len(maps))\n')
    for order in ('hpk_first', 'torch_first'):
        r = subprocess.run([sys.executable, '-c', code, order], cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert r.stdout.decode().split()[-1] == '1', (order, r.stdout.decode())
