#!/usr/bin/env python3
"""C4 compositor only (scripts/bench_configs.py prints the full set)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import bench_configs as B
B.compositor_case(int(sys.argv[1]) if len(sys.argv) > 1 else 30)
