#!/usr/bin/env python3
"""The chain objects round 5 built, at batch scale, for rocprofv3 --kernel-trace --stats: bench.py's cqpsk_p2_chains (P25 Phase 1 CQPSK
chain, P25 Phase 2 chain at 1365 and 4096 channels) and m17_ysf_chains (M17 / YSF at 1365), one JSON object each.
usage: bench_chains.py [cqpsk_p2|m17_ysf|all]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dsd-neo_amd", "bindings"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
import ddn

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("cqpsk_p2", "all"):
    print(json.dumps({"cqpsk_p2_chains": bench.cqpsk_p2_chains(torch, ddn, np, 48000)}))
if which in ("m17_ysf", "all"):
    print(json.dumps({"m17_ysf_chains": bench.m17_ysf_chains(torch, ddn, np, 48000)}))
