"""LAION-UNet leg of bench.py alone (quick iteration)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

torch.cuda.set_device(0)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    print(json.dumps(bench.laion_extras()))
