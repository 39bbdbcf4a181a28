"""Importable alias for the package directory ``carla-social-force-model_amd/``.

The repository layout names the package after the upstream project, hyphens included, which Python
cannot import directly; this shim points the package search path at that directory so that
``import carla_social_force_model_amd.pedestrian_simulation`` resolves to
``carla-social-force-model_amd/pedestrian_simulation.py``.
"""
import os as _os

_impl = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "carla-social-force-model_amd")
if not _os.path.isdir(_impl):  # pragma: no cover
    raise ImportError(f"package directory missing: {_impl}")
__path__ = [_impl]
PACKAGE_DIR = _impl

with open(_os.path.join(_impl, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_impl, "__init__.py"), "exec"))
