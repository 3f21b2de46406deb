"""Regenerates tests/golden/thresholds.json from the CPU oracle (strict IEEE build).
The spot values 38/61/22/18/36/-7 inside it are pinned independently in tests/test_oracle_kat.py
against the reference's own tests (src/test/libIBFTests/read.hpp:154-164)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as po  # noqa: E402

lengths = list(range(0, 2001))
tables = {}
for k in (13, 15):
    for r in (0.1, 0.1 - 0.02):
        tables["%d/%r" % (k, r)] = [po.threshold(L, k, r, 0.95) for L in lengths]
json.dump({"significance": 0.95, "lengths": lengths, "tables": tables},
          open(os.path.join(HERE, "thresholds.json"), "w"))
print("wrote thresholds.json")
