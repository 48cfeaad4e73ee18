"""Times the risk precompute (turbulence index + cov_list) at the reference's panel sizes:
`python bench.py --env riskpre` (one JSON line per shape; kept as a thin wrapper for the evidence scripts)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.exit(subprocess.call([sys.executable, os.path.join(ROOT, "bench.py"), "--env", "riskpre"] + sys.argv[1:]))
