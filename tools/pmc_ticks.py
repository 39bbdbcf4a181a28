"""A few device-resident ticks of one BASELINE workload, whole crowd on one GPU, for rocprofv3 passes:
    rocprofv3 --pmc SQ_INSTS_VALU ... -- python3 tools/pmc_ticks.py c2 40
(the program itself goes after `--`; no launcher in between).  No CPU leg, no collectives."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from carla_social_force_model_amd import scenarios                      # noqa: E402
from carla_social_force_model_amd.config import default_sfm_config      # noqa: E402
from carla_social_force_model_amd.stepper import HipShardEngine         # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sc, forces = scenarios.baseline_scenario(name)
eng = HipShardEngine(default_sfm_config(forces), 0.05)
eng.load(sc)
eng.run(ticks)
eng.synchronize()
meta = os.environ.get("PMC_META")          # pmc_run.sh: where the summariser finds what the last tick's pair kernel evaluated
if meta:
    import json
    try:
        items, terms = eng.engine.pair_work()
    except Exception:
        items, terms = 0, 0
    os.makedirs(os.path.dirname(meta), exist_ok=True)
    with open(meta, "w") as f:
        json.dump({"workload": name, "ticks": ticks, "kernel_variant": eng.engine.kernel_variant(), "pair_items": items, "pair_terms": terms}, f)
eng.close()
