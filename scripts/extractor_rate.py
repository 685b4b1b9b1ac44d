"""Throughput of the ResNet-101 stage-3 feature extractor (bench.py's feature_extraction side object, stand-alone)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import torch
import bench
for n in (16, 64, 128):
    print(json.dumps(bench.extraction_side(torch.device("cuda:0"), n=n)))
