"""bench.py's recurrent_kernel_report alone (the LSTM layer and the two decoders at the headline shapes), for a PMC pass whose
per-kernel counters then belong to exactly these launches (scripts/profile_round.sh -> profiles/*_pmc_recurrent_*)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import json
import torch
import bench

print(json.dumps(bench.recurrent_kernel_report(torch.device("cuda:0"))))
