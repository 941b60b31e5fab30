"""hicpeaks_amd - MI355X (gfx950) HiCCUPS / BH-FDR peak-calling core, drop-in for hicpeaks.callers.

    from hicpeaks_amd.callers import hiccups, bhfdr        # same signatures as hicpeaks 0.3.9

The compute path is libhpk.so (hand-written HIP kernels behind the C ABI of include/hpk.h); importing the
package does not load it, calling a caller does, and fails loudly if it is missing.
"""
__version__ = '0.1.0'
__reference__ = 'hicpeaks 0.3.9'

from ._lib import HpkError, EmptyStepError  # noqa: F401
