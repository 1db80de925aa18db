"""Leaf-module loader for the upstream reference tree (BUILD CONTAINER ONLY).

This file contains no reference source.  It makes a handful of *leaf* modules of
``/root/reference/scratchpad`` importable on CPU so that ``gen_golden.py`` can
run the reference's own Torch / interpreted-Triton code and record input/output
vectors (SURVEY.md section 8c, Appendix A describes the recipe).

It is inert on the GPU box: ``/root/reference`` does not exist there, and nothing
under ``tests/`` imports this module except ``gen_golden.py``.

What it does:
  1. registers ``scratchpad`` as a *namespace* whose ``__path__`` is the reference
     package directory, WITHOUT executing ``scratchpad/utils/__init__.py`` (which
     initialises NVML / zmq / loguru at import time);
  2. provides ``scratchpad.utils`` with only the few names the leaf modules use;
  3. stands in for third-party wheels that are not installed here (flashinfer,
     triteia, zmq, ...) with empty module objects — none of their arithmetic is
     used: every recorded vector comes from the reference's ``forward_native``
     methods or its in-tree Triton kernels run by the Triton interpreter.
"""
import importlib.abc
import importlib.machinery
import importlib.util
import logging
import os
import sys
import types
from unittest.mock import MagicMock

REF_ROOT = os.environ.get("SP_REFERENCE_ROOT", "/root/reference")
REF = os.path.join(REF_ROOT, "scratchpad")

_ABSENT_THIRD_PARTY = [
    "flashinfer", "flashinfer.norm", "flashinfer.activation", "flashinfer.sampling",
    "flashinfer.cascade", "flashinfer.decode",
    "triteia", "triteia.python", "triteia.python.nn", "triteia.python.nn.linear",
    "triteia.triton", "triteia.triton.decode_attention", "triteia.triton.extend_attention",
    "triteia_cuda", "outlines", "xgrammar", "zmq", "loguru", "pynvml", "tenacity",
    "setproctitle", "orjson", "partial_json_parser", "decord", "uvloop",
]


class _Absent(types.ModuleType):
    """Module object for a wheel that is not installed; attributes are MagicMocks."""

    def __getattr__(self, key):
        if key.startswith("__") and key.endswith("__"):
            raise AttributeError(key)
        val = MagicMock(name=f"{self.__name__}.{key}")
        setattr(self, key, val)
        return val


def _install_absent(names):
    for name in names:
        if name in sys.modules:
            continue
        try:
            __import__(name)
            continue
        except Exception:
            pass
        mod = _Absent(name)
        mod.__file__ = f"/nonexistent/{name}.py"
        mod.__path__ = []
        mod.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
        sys.modules[name] = mod
        if "." in name:
            parent, child = name.rsplit(".", 1)
            if parent in sys.modules:
                setattr(sys.modules[parent], child, mod)


class _AbsentSubmoduleFinder(importlib.abc.MetaPathFinder):
    """Resolve submodules of explicitly absent parents only (never a blanket stub)."""

    def find_spec(self, name, path, target=None):
        if "." not in name:
            return None
        if not isinstance(sys.modules.get(name.rsplit(".", 1)[0]), _Absent):
            return None

        class _Loader(importlib.abc.Loader):
            def create_module(self, spec):
                mod = _Absent(spec.name)
                mod.__file__ = "/nonexistent"
                mod.__path__ = []
                return mod

            def exec_module(self, module):
                pass

        return importlib.machinery.ModuleSpec(name, _Loader(), is_package=True)


def _bare_package(name, path, cls=types.ModuleType):
    pkg = cls(name)
    pkg.__path__ = [path]
    pkg.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
    pkg.__spec__.submodule_search_locations = [path]
    sys.modules[name] = pkg
    return pkg


def _load_file(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


_installed = False


def install():
    """Make the reference's leaf modules importable.  Idempotent."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference tree not found at {REF}; fixtures can only be "
                           "regenerated in the build container")
    os.environ.setdefault("TRITON_INTERPRET", "1")
    sys.dont_write_bytecode = True
    import torch

    sys.meta_path.append(_AbsentSubmoduleFinder())
    _install_absent(_ABSENT_THIRD_PARTY)

    pkg = _bare_package("scratchpad", REF)
    utils = _bare_package("scratchpad.utils", REF + "/utils", _Absent)
    pkg.utils = utils
    utils.logger = logging.getLogger("scratchpad")
    lg = types.ModuleType("scratchpad.utils.logger")
    lg.logger = utils.logger
    sys.modules["scratchpad.utils.logger"] = lg
    utils.envs = _load_file("scratchpad.utils.envs", REF + "/utils/envs.py")
    utils.TorchMemorySaverAdapter = _load_file(
        "scratchpad.utils.mem_saver", REF + "/utils/mem_saver.py").TorchMemorySaverAdapter
    utils.current_platform = type(
        "P", (), {"is_cuda_alike": lambda s: False, "is_cuda": lambda s: False})()
    utils.get_compiler_backend = lambda: "inductor"
    utils.supports_custom_op = lambda: hasattr(torch.library, "custom_op")
    _bare_package("scratchpad.server", REF + "/server")
    _bare_package("scratchpad.managers", REF + "/managers").ToppingsManager = MagicMock()
    # module-level capability probes in the in-tree Triton files
    torch.cuda.get_device_capability = lambda *a, **k: (9, 0)
    # MHATokenToKVPool.__init__ asks the device module for a Stream, also on "cpu"
    torch.get_device_module = lambda d=None: types.SimpleNamespace(
        Stream=lambda: None, current_stream=lambda: None)
    _installed = True
