"""Import-path compatibility with the reference's pedestrian_state.py; implementation in host_state.py."""
from .host_state import PED_STATE_DTYPE, PedState  # noqa: F401
