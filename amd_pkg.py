"""Loader for the product package.

The package directory is named ``ts-asr-whisper_amd`` (not a valid Python identifier), so it is
registered under the import name ``ts_asr_whisper_amd``:

    import amd_pkg; pkg = amd_pkg.load()      # afterwards `import ts_asr_whisper_amd.xxx` works
"""
import importlib.util
import os
import sys

NAME = "ts_asr_whisper_amd"
ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "ts-asr-whisper_amd")


def load():
    if NAME in sys.modules:
        return sys.modules[NAME]
    spec = importlib.util.spec_from_file_location(NAME, os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[NAME] = mod
    spec.loader.exec_module(mod)
    return mod
