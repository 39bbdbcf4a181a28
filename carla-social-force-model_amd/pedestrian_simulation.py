"""Import-path compatibility with the reference's pedestrian_simulation.py; implementation in facade.py."""
from .facade import PedestrianSimulation  # noqa: F401
