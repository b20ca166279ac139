"""Import shim: the package directory is ``bias-gan_amd/`` (not a valid Python
identifier), so ``import bias_gan_amd`` loads it from there under this name."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "bias-gan_amd")
_spec = _ilu.spec_from_file_location("bias_gan_amd", _os.path.join(_pkg_dir, "__init__.py"),
                                     submodule_search_locations=[_pkg_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["bias_gan_amd"] = _mod
_spec.loader.exec_module(_mod)
