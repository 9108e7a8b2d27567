"""Shared double-buffered map hand-off (replacement for clone_obj + mp.Queue, utils/multiprocessing_utils.py:21-31)."""
import torch
import torch.multiprocessing as mp

from monogs_amd.map_arena import MapArena

FIELDS = {"xyz": (3,), "rgb": (3,), "opacity": (1,), "scaling": (1,), "rotation": (4,)}


def _map(n, seed):
    g = torch.Generator().manual_seed(seed)
    return {k: torch.randn(n, *s, generator=g) for k, s in FIELDS.items()}


def test_publish_acquire_double_buffer():
    a = MapArena(1000, FIELDS)
    assert a.acquire() == (0, {})
    m1, m2, m3 = _map(300, 1), _map(450, 2), _map(20, 3)
    assert a.publish(m1) == 1
    s1, v1 = a.acquire()
    assert s1 == 1 and all(torch.equal(v1[k], m1[k]) for k in FIELDS)
    assert a.publish(m2) == 2
    s2, v2 = a.acquire()
    assert s2 == 2 and v2["xyz"].shape[0] == 450 and torch.equal(v2["rotation"], m2["rotation"])
    assert torch.equal(v1["xyz"], m1["xyz"]) and not a.stale(s1)       # the first snapshot is still intact
    a.publish(m3)
    assert a.stale(s1) and not a.stale(s2)
    assert torch.equal(v2["rgb"], m2["rgb"])                            # slot of s2 untouched by the third publish
    assert v1["xyz"].data_ptr() == a.acquire()[1]["xyz"].data_ptr()     # only two buffers, no allocation per publish


def test_errors():
    a = MapArena(10, FIELDS)
    import pytest
    with pytest.raises(ValueError):
        a.publish(_map(11, 0))
    bad = _map(5, 0)
    bad.pop("rgb")
    with pytest.raises(ValueError):
        a.publish(bad)


def _reader(arena, out_q, n_expected):
    seen = 0
    while seen < n_expected:
        seq, views = arena.acquire()
        if seq > seen:
            # the content encodes its own sequence: every entry of xyz equals seq
            ok = bool((views["xyz"] == float(seq)).all()) and views["xyz"].shape[0] == 10 * seq
            out_q.put((seq, ok))
            seen = seq
    out_q.put(("done", True))


def test_two_processes_share_the_buffers():
    ctx = mp.get_context("spawn")
    arena = MapArena(200, FIELDS)
    q = ctx.Queue()
    n = 5
    p = ctx.Process(target=_reader, args=(arena, q, n))
    p.start()
    import time
    for seq in range(1, n + 1):
        m = {k: torch.full((10 * seq, *s), float(seq)) for k, s in FIELDS.items()}
        arena.publish(m)
        time.sleep(0.2)                    # let the reader see every sequence (it polls)
    results = []
    while True:
        item = q.get(timeout=60)
        if item[0] == "done":
            break
        results.append(item)
    p.join(timeout=30)
    assert p.exitcode == 0
    assert results and all(ok for _, ok in results) and results[-1][0] == n
