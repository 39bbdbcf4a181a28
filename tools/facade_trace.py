"""A few hundred drop-in ticks at N = 20 for `rocprofv3 --kernel-trace --stats` (which kernels a drop-in tick is made of, how long each runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.ped_mode_manager import PedMode, PedModeManager
from carla_social_force_model_amd.pedestrian_simulation import PedestrianSimulation
n, nb, ns, nd = (int(sys.argv[1]) if len(sys.argv) > 1 else 20), 8, 4, 2
sc = scenarios.make_scenario(n, 1, nb, ns, nd)
info = [[sc.border_centers[k], float(sc.border_lengths[k])] for k in range(nb)]
sim = PedestrianSimulation(sc.borders, info, sc.static_obstacles, default_sfm_config(), 0.05, record_states=False)
for i in range(n):
    nm = f"ped_{i}"
    sim.spawn_pedestrian((nm, i, sc.loc[i], sc.vel[i], sc.waypoint[i], PedModeManager(nm, 1.2, PedMode.WALKING_SIDEWALK, 1.5, 1.5), 0.3, 1.2))
dyn = (list(range(nd)), [c for c, _ in sc.dynamic_obstacles], [0.0] * nd, list(sc.dynamic_vel), [np.array([2.4, 1.0])] * nd, [r for _, r in sc.dynamic_obstacles])
for k in range(400):
    sim.update_dynamic_obstacles(dyn); sim.tick(0.05 * k); sim.get_new_velocities()
sim.close()
