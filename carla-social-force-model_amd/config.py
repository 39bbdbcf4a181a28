"""The sfm_config.toml parameter surface.

``default_sfm_config()`` returns the stock parameter values of the reference's config/sfm_config.toml
as a plain dict; ``load_sfm_config(path)`` reads a TOML file the way run_simulation.py:232-240 does
(tomli, binary mode).  Keys are consumed bug-compatibly by ``params.SfmParams.from_config``.
"""
from __future__ import annotations

import copy
import os

_STOCK = {
    "max_speed_multiplier": 1.3,   # present in the stock file but never read (pedestrian_state.py:15)
    "use_ped_radius": False,
    "forces": {"acceleration_force": True, "pedestrian_force": True, "border_force": True,
               "static_obstacle_force": True, "dynamic_obstacle_force": True},
    "acceleration_force": {"tau": 0.5},   # never read either (forces.py:44 looks under goal_force)
    "pedestrian_force": {"lambda": 2.0, "A": 4.5, "gamma": 0.35, "n": 2.0, "n_prime": 3.0, "epsilon": 0.005},
    "border_force": {"a": 6.0, "b": 0.3},
    "static_obstacle_force": {"lambda": 2.3, "A": 15, "gamma": 0.4, "n": 2.1, "n_prime": 3.0,
                              "epsilon": 0.005, "perception_threshold": 20},
    "dynamic_obstacle_force": {"lambda": 2.0, "A": 50, "gamma": 0.4, "n": 1.0, "n_prime": 3.0,
                               "epsilon": 0.005, "perception_threshold": 50},
}

DEFAULT_TOML = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config", "sfm_config.toml")


def default_sfm_config(forces=None):
    """Stock parameters; ``forces`` (iterable of names) switches exactly those forces on."""
    cfg = copy.deepcopy(_STOCK)
    if forces is not None:
        cfg["forces"] = {k: (k in forces) for k in _STOCK["forces"]}
    return cfg


def load_sfm_config(path=DEFAULT_TOML):
    import tomli
    with open(path, "rb") as f:
        return tomli.load(f)
