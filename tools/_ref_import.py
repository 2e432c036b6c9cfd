"""Import helper for the golden-vector generators (runs ONLY in the build container).

The reference (`/root/reference`, judo v0.0.7) is Python; its numpy/scipy code for
the hot path (optimizers, spline, rewards, normalisers, `Controller.update_action`)
is importable here once the third-party packages that are absent from this image
(`mujoco`, `viser`, `trimesh`, `omegaconf`, `dora_utils`, `mujoco_extensions`,
`onnxruntime`, ...) are replaced by inert stub modules.  Nothing in this file or in
the reference travels to the GPU box: the generators write plain `.npz` data into
`tests/golden/` and that is the only thing the test-suite reads.
"""

from __future__ import annotations

import importlib.abc
import importlib.machinery
import sys
import types

REFERENCE_ROOT = "/root/reference"

_MISSING_TOPLEVEL = (
    "mujoco",
    "viser",
    "trimesh",
    "omegaconf",
    "dora_utils",
    "dora",
    "mujoco_extensions",
    "onnxruntime",
    "robot_descriptions",
    "hydra",
    "tyro",
    "rich",
)


class _StubModule(types.ModuleType):
    """A module whose every attribute is a fresh empty class (enough for `from x import Y`)."""

    def __getattr__(self, name: str):
        if name.startswith("__"):
            raise AttributeError(name)
        obj = type(name, (), {})
        setattr(self, name, obj)
        return obj


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        top = fullname.split(".")[0]
        if top in _MISSING_TOPLEVEL:
            try:  # use the real one if it ever becomes available
                for finder in sys.meta_path:
                    if finder is self:
                        continue
                    spec = finder.find_spec(fullname, path, target) if hasattr(finder, "find_spec") else None
                    if spec is not None:
                        return spec
            except Exception:
                pass
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        mod = _StubModule(spec.name)
        mod.__path__ = []  # behaves like a package so that submodule imports resolve
        return mod

    def exec_module(self, module):
        if module.__name__ == "mujoco":
            module.mj_forward = lambda *a, **k: None


def install() -> None:
    """Install the stubs and put the reference on `sys.path`."""
    if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.append(_StubFinder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
