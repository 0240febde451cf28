"""Imports the data tables of the reference that the hot path needs verbatim (run here, where /root/reference is mounted):

  ai-toolkit_amd/data/flowmatch_default_weighing_scheme.json
      the 1000 per-timestep loss weights of `timestep_type: weighted` (toolkit/timestep_weighing/default_weighing_scheme.py, measured
      by the reference's authors on flex.1-alpha; consumed at toolkit/samplers/custom_flowmatch_sampler.py:59-76).  Empirical data, not
      derivable — result parity requires the same numbers.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
from toolkit.timestep_weighing.default_weighing_scheme import default_weighing_scheme  # noqa: E402

out = os.path.join(ROOT, "ai-toolkit_amd", "data", "flowmatch_default_weighing_scheme.json")
assert len(default_weighing_scheme) == 1000
json.dump({"source": "ostris/ai-toolkit toolkit/timestep_weighing/default_weighing_scheme.py", "weights": [float(x) for x in default_weighing_scheme]},
          open(out, "w"))
print("wrote", out)
