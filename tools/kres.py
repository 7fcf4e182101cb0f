"""tools/kres.py [src=hr_render.hip] [filter] — registers / occupancy / scratch of every kernel of a source, from the compiler's remarks (no GPU)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from test_kernel_resources import _resources
src = sys.argv[1] if len(sys.argv) > 1 else "hr_render.hip"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, v in _resources(src).items():
    if flt in k:
        print(f"{k.split('(')[0][:70]:70s} VGPR {v.get('VGPRs')} AGPR {v.get('AGPRs')} SGPR {v.get('SGPRs')} occ {v.get('Occupancy')} scratch {v.get('ScratchSize')} spill {v.get('VGPRs Spill')} LDS {v.get('LDS Size')}")
