"""Import-path compatibility with the reference's ped_mode_manager.py; implementation in host_state.py."""
from .host_state import BORDER_FREE_MODES, PedMode, PedModeManager  # noqa: F401
