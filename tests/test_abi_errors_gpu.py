"""GPU tests of the C ABI's error behaviour: bad arguments and call-order mistakes return a negative status with a
message (no exceptions cross the ABI, nothing falls back to a CPU path)."""
import ctypes as C

import numpy as np
import pytest

from carla_social_force_model_amd import _lib
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.engine import SfmEngine, params_from_config

pytestmark = pytest.mark.gpu


def _err(lib, h):
    return lib.sfm_last_error(h).decode()


def test_argument_and_state_errors():
    lib = _lib.load()
    p = params_from_config(default_sfm_config(), 0.05)
    h = C.c_void_p()
    assert lib.sfm_create(None, 0, C.byref(h)) == -1 and "NULL" in _err(lib, None)
    assert lib.sfm_create(C.byref(p), 999, C.byref(h)) == -1 and "device_id" in _err(lib, None)
    bad = params_from_config(default_sfm_config(), 0.05)
    bad.step_length = 0.0
    assert lib.sfm_create(C.byref(bad), 0, C.byref(h)) == -1 and "step_length" in _err(lib, None)
    assert lib.sfm_create(C.byref(p), 0, C.byref(h)) == 0
    try:
        assert lib.sfm_tick(h, 0) == 0                                        # no pedestrians: a no-op (pedestrian_simulation.py:60-61)
        one = np.zeros(1, np.float32)
        f = one.ctypes.data_as(_lib._F)
        assert lib.sfm_upload_state(h, -1, f, f, None, f, f, None, f, f, f, None, None) == -1
        assert lib.sfm_upload_state(h, 1, None, f, None, f, f, None, f, f, f, None, None) == -1 and "NULL" in _err(lib, h)
        assert lib.sfm_upload_state(h, 1, f, f, f, f, f, None, f, f, f, None, None) == -1 and "together" in _err(lib, h)
        assert lib.sfm_upload_state(h, 1, f, f, None, f, f, None, f, f, f, None, None) == 0
        assert lib.sfm_download_forces(h, 1, f, f, None) == -3 and "RECORD" in _err(lib, h)        # no recorded tick yet
        assert lib.sfm_download_forces(h, 9, f, f, None) == -1
        assert lib.sfm_set_shard(h, 0, 5) == -1 and "range" in _err(lib, h)
        off = np.array([0, 3, 2], np.int32)
        assert lib.sfm_set_borders(h, 2, off.ctypes.data_as(_lib._I), f, f, f, f, f) == -1 and "non-decreasing" in _err(lib, h)
        assert lib.sfm_set_borders(h, -1, None, None, None, None, None, None) == -1
        assert lib.sfm_get_timing(h, None, None, None) == 0 or True
        assert lib.sfm_download_modes(h, None, None, None) == -3                                  # FSM not set
        assert lib.sfm_run_recorded(h, 5, 0, 0, None, 0, None) == -1
        assert lib.sfm_tick(h, 4) == 0 and lib.sfm_download_forces(h, 5, f, f, None) == 0
        # round-2 entry points
        bnd = np.array([0, 64, 32], np.int32)
        assert lib.sfm_set_partition(h, 2, 1, bnd.ctypes.data_as(_lib._I)) == -1 and "non-decreasing" in _err(lib, h)
        assert lib.sfm_set_partition(h, 5, 5, None) == -1 and "16" in _err(lib, h)
        assert lib.sfm_set_partition(h, 2, 0, None) == -1
        assert lib.sfm_set_partition(h, 0, 0, None) == 0
        items, terms = C.c_longlong(0), C.c_longlong(0)
        assert lib.sfm_get_pair_work(h, C.byref(items), C.byref(terms)) == -3 and "symmetric" in _err(lib, h)   # one pedestrian: ordered kernel
        assert lib.sfm_tick_begin(h, 0) == 0 and lib.sfm_tick_end(h, 0) == 0                      # not a shard: begin is a no-op, end the whole tick
        ms, tk, ln = C.c_float(0), C.c_int(0), C.c_int(0)
        assert lib.sfm_get_timing(h, C.byref(ms), C.byref(tk), C.byref(ln)) == 0 and tk.value == 1 and ln.value >= 1
        assert lib.sfm_set_timing(h, 0) == 0 and lib.sfm_tick(h, 0) == 0                          # the event bracket off: ticks run, ...
        assert lib.sfm_get_timing(h, C.byref(ms), C.byref(tk), C.byref(ln)) == -3 and "sfm_set_timing" in _err(lib, h)   # ... nothing to report
        assert lib.sfm_set_timing(h, 1) == 0 and lib.sfm_tick(h, 0) == 0 and lib.sfm_get_timing(h, C.byref(ms), C.byref(tk), C.byref(ln)) == 0
        assert lib.sfm_set_timing(None, 1) == -1
    finally:
        assert lib.sfm_destroy(h) == 0
    assert lib.sfm_destroy(None) == 0


def test_use_ped_radius_requires_radius_and_key_errors_surface():
    cfg = default_sfm_config()
    cfg["use_ped_radius"] = True
    eng = SfmEngine(cfg, 0.05)
    try:
        z = np.zeros((3, 3))
        with pytest.raises(_lib.SfmLibraryError, match="radius"):
            eng.upload_state(z, z, z, np.ones(3), None, None)
    finally:
        eng.close()
    bad = default_sfm_config()
    del bad["border_force"]
    with pytest.raises(KeyError):                                             # forces.py:134 indexes the table
        SfmEngine(bad, 0.05)
