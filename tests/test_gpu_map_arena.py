"""MapArena with device buffers shared between two processes (HIP IPC through torch.multiprocessing)."""
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
FIELDS = {"xyz": (3,), "rotation": (4,)}


def _reader(arena, q):
    import time
    t0 = time.time()
    while time.time() - t0 < 60:
        seq, views = arena.acquire()
        if seq >= 2:
            q.put((seq, int(views["xyz"].shape[0]), float(views["xyz"].sum().item()), str(views["xyz"].device)))
            # drop every IPC mapping before this process ends (otherwise the producer's allocator warns that a
            # consumer died holding shared device tensors)
            # (the Process object keeps the unpickled arguments alive and a spawned child leaves through os._exit, which runs
            #  no destructors: drop that reference too)
            import multiprocessing
            multiprocessing.current_process()._args = ()
            del views, arena
            import gc
            gc.collect()
            torch.cuda.ipc_collect()
            return
        time.sleep(0.01)
    q.put(("timeout",))


def test_device_buffers_across_processes():
    from monogs_amd.map_arena import MapArena
    ctx = mp.get_context("spawn")
    arena = MapArena(1000, FIELDS, device="cuda:0")
    arena.publish({"xyz": torch.ones(100, 3, device="cuda:0"), "rotation": torch.zeros(100, 4, device="cuda:0")})
    q = ctx.Queue()
    p = ctx.Process(target=_reader, args=(arena, q))
    p.start()
    arena.publish({"xyz": torch.full((250, 3), 2.0, device="cuda:0"), "rotation": torch.zeros(250, 4, device="cuda:0")})
    got = q.get(timeout=120)
    p.join(timeout=30)
    assert got == (2, 250, 1500.0, "cuda:0"), got
    del arena                      # release the exported IPC blocks before this (producer) process goes on
    import gc
    import time
    for _ in range(5):             # (the consumer's reference counts reach the producer's limbo list a moment after it exits)
        gc.collect()
        torch.cuda.ipc_collect()
        time.sleep(0.05)
