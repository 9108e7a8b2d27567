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


def test_reader_holding_views_across_two_publishes_sees_intact_data_or_stale():
    """A reader that keeps its views while the writer publishes twice more reads either an intact snapshot or gets
    ``stale() == True`` when it re-checks after reading -- never a torn map that passes for valid.  Every element of a
    published map equals its sequence number, so a torn read shows up as a non-constant tensor."""
    import threading
    n = 400_000
    a = MapArena(n, {"xyz": (3,), "rotation": (4,)})
    a.publish({"xyz": torch.full((n, 3), 1.0), "rotation": torch.full((n, 4), 1.0)})
    stop = threading.Event()
    published = [1]

    def writer():
        s = 1
        while not stop.is_set():
            s += 1
            a.publish({"xyz": torch.full((n, 3), float(s)), "rotation": torch.full((n, 4), float(s))})
            published[0] = s

    t = threading.Thread(target=writer)
    t.start()
    torn_but_valid, valid_reads, stale_reads = 0, 0, 0
    try:
        import time
        t_end = time.time() + 3.0
        while time.time() < t_end:
            seq, views = a.acquire()
            time.sleep(0.002)                         # hold the views while the writer keeps publishing
            lo = min(float(v.min()) for v in views.values())
            hi = max(float(v.max()) for v in views.values())
            if a.stale(seq):                          # re-check AFTER reading: the data may be torn, discard it
                stale_reads += 1
                continue
            valid_reads += 1
            if not (lo == hi == float(seq)):
                torn_but_valid += 1
    finally:
        stop.set()
        t.join()
    assert published[0] > 10, "the writer did not race the reader"
    assert torn_but_valid == 0, (torn_but_valid, valid_reads, stale_reads)
    assert stale_reads > 0                            # the race really happened


def test_stale_is_raised_while_the_readers_slot_is_being_overwritten():
    """Deterministic version of the race: between 'write in progress' and the header update of the publish after next,
    ``stale()`` of the reader's sequence is already True (it used to be False for the whole copy)."""
    a = MapArena(8, {"x": (1,)})
    a.publish({"x": torch.zeros(8, 1)})
    seq, _ = a.acquire()
    a.publish({"x": torch.ones(8, 1)})
    assert not a.stale(seq)                           # the other slot was written: the reader's slot is intact
    a.header[3] = (int(a.header[1]) ^ 1) + 1            # what publish() does first: mark the reader's slot as being written
    assert a.stale(seq)
    a.header[3] = 0
    assert not a.stale(seq)
