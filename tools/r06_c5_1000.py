#!/usr/bin/env python3
"""bench.py's fragsplit_config5_1000 object alone (1000 files x 1e4 fragments through the fused fragment pipeline)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
torch.cuda.set_device(0)
print(json.dumps(bench.bench_fragsplit_many_files()))
