#!/usr/bin/env python3
"""Map hand-off latency between two processes on one GPU (mapper -> tracker), SURVEY.md section 8f rank 4:

  arena   monogs_amd.map_arena.MapArena: attach once, device-to-device copy into the back buffer, header publish;
  queue   what the reference does on every keyframe (/root/reference/utils/slam_mapper.py:550-564 with
          /root/reference/utils/multiprocessing_utils.py:21-31): deep-copy + clone every tensor of the map object and
          send it through a torch.multiprocessing Queue (pickling = one HIP-IPC handle export per tensor, and an
          import + mapping on the receiving side).

For each hand-off the producer stamps a sequence number into the map, publishes, and the consumer reports when it can
read that number from device memory (perf_counter is the machine-wide monotonic clock).  Prints one JSON line.
  python tools/handoff_bench.py [--gaussians 100000] [--rounds 30]
"""
import argparse
import copy
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.multiprocessing as mp

FIELDS = {"xyz": (3,), "f_dc": (3,), "opacity": (1,), "scaling": (1,), "rotation": (4,)}


class MapObject:                                  # stand-in for GaussianModel: the tensors clone_obj() copies
    def __init__(self, n, dev):
        for k, s in FIELDS.items():
            setattr(self, "_" + k, torch.zeros(n, *s, device=dev))
        self.max_radii_2d = torch.zeros(n, device=dev)
        self.xyz_gradient_accum = torch.zeros(n, 1, device=dev)
        self.denom = torch.zeros(n, 1, device=dev)
        self.unique_kfIDs = torch.zeros(n, dtype=torch.int32)
        self.n_obs = torch.zeros(n, dtype=torch.int32)


def clone_obj(obj):                               # same recipe as the reference's helper
    c = copy.deepcopy(obj)
    for attr in c.__dict__.keys():
        if isinstance(getattr(c, attr), torch.Tensor):
            setattr(c, attr, getattr(c, attr).detach().clone())
    return c


def consumer_arena(arena, rounds, q_out, ready):
    torch.cuda.set_device(0)
    ready.set()
    seen, stamps, last = 0, [], None
    deadline = time.time() + 30
    while len(stamps) < rounds and time.time() < deadline:
        seq, views = arena.acquire()
        if seq > seen:
            v = float(views["opacity"][0, 0].item())          # the data is readable from device memory
            last = (seq, v)
            if not arena.stale(seq) and int(v) == seq:
                stamps.append((seq, time.perf_counter()))
                seen = seq
        else:
            time.sleep(0.0002)
    q_out.put((stamps, last))
    views = None
    del arena
    torch.cuda.ipc_collect()


def consumer_queue(q_in, rounds, q_out, ready):
    torch.cuda.set_device(0)
    ready.set()
    stamps = []
    try:
        for _ in range(rounds):
            tag, gm, n_kf = q_in.get(timeout=30)
            seq = int(gm._opacity[0, 0].item())
            stamps.append((seq, time.perf_counter()))
            del gm
    except Exception as e:          # report what arrived
        print("[consumer_queue]", repr(e), file=sys.stderr, flush=True)
    q_out.put((stamps, None))
    torch.cuda.ipc_collect()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=100_000)
    ap.add_argument("--rounds", type=int, default=30)
    a = ap.parse_args()
    dev = "cuda:0"
    ctx = mp.get_context("spawn")
    from monogs_amd.map_arena import MapArena
    n, R = a.gaussians, a.rounds
    live = MapObject(n, dev)
    out = {"gaussians": n, "rounds": R, "bytes_per_handoff": 4 * 12 * n}

    # ---- arena
    arena = MapArena(n, FIELDS, device=dev)
    q_out, ready = ctx.Queue(), ctx.Event()
    p = ctx.Process(target=consumer_arena, args=(arena, R, q_out, ready))
    p.start(); ready.wait(120); time.sleep(1.0)
    sends, pub_ms = {}, []
    for s in range(1, R + 1):
        live._opacity.fill_(float(s)); torch.cuda.synchronize()
        t0 = time.perf_counter()
        arena.publish({k: getattr(live, "_" + k) for k in FIELDS})
        pub_ms.append((time.perf_counter() - t0) * 1e3)
        sends[s] = t0
        time.sleep(0.02)
    stamps, last = q_out.get(timeout=60); p.join(30)
    lat = sorted((t - sends[s]) * 1e3 for s, t in stamps) or [float("nan")]
    out["arena"] = {"received": len(stamps), "last_seen": last, "producer_ms_median": round(sorted(pub_ms)[R // 2], 3), "end_to_end_ms_median": round(lat[len(lat) // 2], 3),
                    "end_to_end_ms_max": round(lat[-1], 3)}

    # ---- clone_obj + Queue
    q_in, q_out, ready = ctx.Queue(), ctx.Queue(), ctx.Event()
    p = ctx.Process(target=consumer_queue, args=(q_in, R, q_out, ready))
    p.start(); ready.wait(120); time.sleep(1.0)
    sends, pub_ms = {}, []
    for s in range(1, R + 1):
        live._opacity.fill_(float(s)); torch.cuda.synchronize()
        t0 = time.perf_counter()
        q_in.put(["sync_backend", clone_obj(live), 8])
        pub_ms.append((time.perf_counter() - t0) * 1e3)
        sends[s] = t0
        time.sleep(0.02)
    stamps, _ = q_out.get(timeout=90); p.join(30)
    lat = sorted((t - sends[s]) * 1e3 for s, t in stamps) or [float("nan")]
    out["clone_obj_queue"] = {"received": len(stamps), "producer_ms_median": round(sorted(pub_ms)[R // 2], 3),
                              "end_to_end_ms_median": round(lat[len(lat) // 2], 3), "end_to_end_ms_max": round(lat[-1], 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
