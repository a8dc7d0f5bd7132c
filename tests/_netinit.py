"""Test-side alias of network/netinit.py (deterministic name-keyed weights, the cfg dict, seeded synthetic frames): the
helpers live with the product's entry points because bench.py and __graft_entry__.smoke() use them too."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "network"))
from netinit import deterministic_init, make_cfg, synthetic_frames  # noqa: E402,F401
