"""MI355X-native Social Force Model stepper behind the reference's PedestrianSimulation / Force API.

Hot path only: forces.py + stateutils.py + the numeric half of
pedestrian_simulation.py of felixlutz/carla-social-force-model, as hand-written HIP kernels for gfx950
called through the C ABI in ``include/sfm_hip.h``.  Import as ``carla_social_force_model_amd``.
"""
__version__ = "0.1.0"
