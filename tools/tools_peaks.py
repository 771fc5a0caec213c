"""Prints bench.py's measured-attainable peaks of this device (float4 copy, fp16 MFMA loops on constant / changing operands)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(json.dumps(bench.measure_peaks(torch.device("cuda")), indent=1))
