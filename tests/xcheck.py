"""Test infrastructure: the cross-check kernel generations (tests/libjudo_amd_xcheck.so, built by `__graft_entry__.build_xcheck`).

The product library ships one kernel generation per model.  The parity suite also compares it with two older / independent GPU implementations of the same
step (generation 1: one lane per rollout, model-generic; generation 2: the cooperative kernels of rounds 1 and 2); those kernels live in a test-only shared
library that hands its launchers to the product library through `jh_register_xcheck` (include/judo_amd.h).  `load()` is idempotent."""

from __future__ import annotations

import ctypes as C
import os

_X: C.CDLL | None = None
PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libjudo_amd_xcheck.so")


def load() -> C.CDLL:
    global _X
    if _X is None:
        from judo_amd import _lib

        _lib.lib()  # the product library first: the cross-check library links against it and must bind to the instance already in the process
        if not os.path.exists(PATH):
            raise ImportError(f"{PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        X = C.CDLL(PATH)
        X.jh_xcheck_register.restype = C.c_int
        _lib.check(X.jh_xcheck_register(), "jh_xcheck_register")
        _X = X
    return _X
