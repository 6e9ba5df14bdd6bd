#!/usr/bin/env python3
"""Launcher for super-resolution_amd.esrgan (the package directory name has a hyphen, so ``-m`` cannot name it).

    python tools/train.py --residual_blocks 23 --factor 4 --hr_height 256 --hr_width 256 --batch_size 32 --n_batches 100
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train.py ...   # data parallel
"""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
es = importlib.import_module("super-resolution_amd.esrgan")
if __name__ == "__main__":
    info = es.train(es.get_parser())
    if int(os.environ.get("RANK", 0)) == 0:
        print(json.dumps({k: v for k, v in info.items() if k not in ("loss", "argument")}, default=str))
