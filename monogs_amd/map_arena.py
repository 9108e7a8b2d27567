"""Double-buffered shared arena for handing the Gaussian map from the mapper to the tracker / viewer
(SURVEY.md section 8f rank 4).

The reference deep-copies the whole ``GaussianModel`` (``clone_obj``: ``copy.deepcopy`` + a ``clone()`` of every
tensor, /root/reference/utils/multiprocessing_utils.py:21-31) and pickles it through an ``mp.Queue`` on every keyframe
(/root/reference/utils/slam_mapper.py:550-564): an O(P) copy, an allocation and an IPC-handle exchange per hand-off.

Here both sides attach ONCE to two pre-allocated buffers per parameter (``torch`` shared memory for CPU tensors, CUDA/HIP
IPC for device tensors -- ``torch.multiprocessing`` shares both the same way) and a small shared header:

* the writer copies the live map into the buffer that is NOT published (one device-to-device copy per parameter, no
  allocation) and then publishes it by storing ``(sequence, slot, count)`` in the header;
* a reader takes a consistent snapshot with a seqlock: read the header, take views of ``[:count]`` of that slot, read
  the header again -- if the sequence moved, retry.  The writer never touches the PUBLISHED slot, but with two slots
  the publish after next re-uses the reader's slot, and it starts copying into it before the header shows the new
  sequence.  The header therefore carries a write-in-progress word that the writer sets BEFORE the first byte is
  copied: a reader that took sequence ``s`` may use its views while ``stale(s)`` is False, and must re-check
  ``stale(s)`` AFTER it has finished reading them (seqlock read side) -- True means the data may be torn: discard
  and ``acquire()`` again.

Capacity is fixed at creation (MonoGS maps grow by at most a few thousand Gaussians per keyframe; size it for the
session, 288 GB of HBM is not the constraint).  Works with any set of named tensors, so the activated tensors the tracker
renders from can ride along with the raw parameters.
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import torch


class MapArena:
    HEADER = 4           # int64: [sequence, slot, count, write in progress (1 + target slot, 0 = none)]

    def __init__(self, capacity: int, fields: Dict[str, Sequence[int]], device="cpu", dtype=torch.float32):
        """``fields`` maps a name to the trailing shape of one Gaussian's entry, e.g. {"xyz": (3,), "rotation": (4,)}."""
        self.capacity = int(capacity)
        self.fields = {k: tuple(v) for k, v in fields.items()}
        self.buffers = {k: [torch.zeros((self.capacity, *shape), dtype=dtype, device=device) for _ in range(2)]
                        for k, shape in self.fields.items()}
        self.header = torch.zeros(self.HEADER, dtype=torch.int64)          # always host memory: tiny and polled
        self.share_memory_()

    def share_memory_(self):
        self.header.share_memory_()
        for pair in self.buffers.values():
            for t in pair:
                t.share_memory_()          # no-op for device tensors: torch.multiprocessing sends them as IPC handles
        return self

    # ---- writer (mapper) ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def publish(self, tensors: Dict[str, torch.Tensor]) -> int:
        """Copy ``tensors`` (same keys as ``fields``, same length P <= capacity) into the back buffer and publish it.
        Returns the new sequence number."""
        seq, slot = int(self.header[0]), int(self.header[1])
        back = 1 - slot if seq > 0 else 0
        counts = {int(t.shape[0]) for t in tensors.values()}
        if set(tensors) != set(self.fields) or len(counts) != 1:
            raise ValueError("publish() needs exactly the arena's fields, all of the same length")
        n = counts.pop()
        if n > self.capacity:
            raise ValueError(f"map has {n} Gaussians, arena capacity is {self.capacity}")
        self.header[3] = back + 1          # write in progress: readers still holding this slot (sequence seq - 1) are stale NOW
        for k, t in tensors.items():
            self.buffers[k][back][:n].copy_(t.detach().reshape(n, *self.fields[k]))
        dev = next(iter(self.buffers.values()))[0].device
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()      # the copies are complete before the header says so
        # count and slot first, sequence last: a reader that sees the new sequence sees its slot and count
        self.header[2] = n
        self.header[1] = back
        self.header[0] = seq + 1
        self.header[3] = 0
        return seq + 1

    # ---- readers (tracker, viewer) --------------------------------------------------------------------------------
    def acquire(self, max_retries: int = 1000) -> Tuple[int, Dict[str, torch.Tensor]]:
        """(sequence, {name: view of the published [count, ...] tensor}); sequence 0 = nothing published yet."""
        for _ in range(max_retries):
            seq = int(self.header[0])
            if seq == 0:
                return 0, {}
            slot, n = int(self.header[1]), int(self.header[2])
            views = {k: pair[slot][:n] for k, pair in self.buffers.items()}
            if int(self.header[0]) == seq:
                return seq, views
        raise RuntimeError("MapArena.acquire: the writer kept publishing; no consistent header")

    def stale(self, seq: int) -> bool:
        """True once the writer may have touched the slot a reader took at ``seq``: the sequence is two ahead, or it is
        one ahead and a write is in progress (with two slots that write targets the reader's slot).  Check it after
        reading the views, not only before."""
        writing = int(self.header[3])       # read before the sequence: a write that completes in between shows as seq + 2
        cur = int(self.header[0])
        return cur >= seq + 2 or (cur == seq + 1 and writing != 0)
