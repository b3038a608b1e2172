# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe (run under tools/trace_kernels.sh): scans with 259 / 65899 candidates on 4 GiB."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
mm = load_package()
eng = mm.Engine(0)
n = 4 << 30
eng.alloc(n); eng.synth(42)
for kw in ("cake", "the"):
    plan = mm.plan_relative(1, kw)
    for i in range(60):
        eng.scan(plan, block_bytes=524288, cap=1 << 18)
