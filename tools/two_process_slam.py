#!/usr/bin/env python3
"""Tracker (this process) + mapper (a spawned process) on one GPU, the map handed over through MapArena -- the process
topology of /root/reference/slam.py:102-179 on the synthetic TUM-like sequence.  Prints one JSON line.
  python tools/two_process_slam.py [--frames 21] [--eager]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=21)
    ap.add_argument("--eager", action="store_true", help="mapper without hipGraph replay")
    a = ap.parse_args()
    from monogs_amd.slam_harness import run_slam_two_process
    r = run_slam_two_process(n_frames=a.frames, intrinsics="fr3_office", tracking_itr_num=100, mapping_itr_num=150,
                             window_size=8, kf_interval=5, init_itr_num=150, graph=not a.eager)
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}))
