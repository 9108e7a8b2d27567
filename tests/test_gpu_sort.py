"""The hand-written radix sort on its own (pytest -m gpu), through the C ABI's test entry mgs_debug_sort_pairs.

What it replaces is cub::DeviceRadixSort::SortPairs (/root/reference's rasteriser submodule calls
it on 64-bit (tile | depth) keys): a STABLE sort.  The library ranks a pair inside its wave with one returning LDS atomic
per item; that is stable only if the LDS unit applies the lanes of one DS instruction to one address in ascending lane
order, which is what these cases pin: keys with one to five distinct digits (every wave-instruction full of equal
digits) must come out exactly as torch's stable sort and as the ballot ranking (the first implementation, kept as a
test knob) put them, on every path (1024 / 2048-pair tiles with the all-gather of counts, 3072 / 4096-pair counted tiles)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sort(lib, keys, bits):
    from monogs_amd._lib import check
    n = keys.numel()
    k = keys.clone()
    v = torch.arange(n, device=DEV, dtype=torch.int32)
    ka, va = torch.empty_like(k), torch.empty_like(v)
    temp = torch.empty(lib.mgs_debug_sort_temp_bytes(n, bits), dtype=torch.uint8, device=DEV)
    check(lib.mgs_debug_sort_pairs(k.data_ptr(), v.data_ptr(), ka.data_ptr(), va.data_ptr(), n, bits, temp.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream), "mgs_debug_sort_pairs")
    return k, v


def _keys(kind, n, bits, gen):
    top = (1 << bits) - 1
    if kind == "uniform":
        return torch.randint(0, top + 1, (n,), generator=gen, dtype=torch.int64)
    if kind == "five":          # five distinct keys, two digits each: the adversarial case for the ranking
        pick = torch.tensor([0, 1, 2, 257, 258]) & top
        return pick[torch.randint(0, 5, (n,), generator=gen)]
    if kind == "one":           # every key equal: the sort must be the identity
        return torch.full((n,), 0x5A5A5A5A & top, dtype=torch.int64)
    if kind == "sorted":
        return (torch.arange(n, dtype=torch.int64) * 7919) & top if bits < 20 else torch.arange(n, dtype=torch.int64) & top
    if kind == "depth":         # float bits of depths in [0.2, 12), a quarter culled (all ones): the forward's first sort
        z = torch.rand(n, generator=gen) * 11.8 + 0.2
        k = z.view(torch.int32).long()
        k[torch.rand(n, generator=gen) < 0.25] = 0xFFFFFFFF
        return k & top
    raise ValueError(kind)


# sizes that take every kernel configuration: 1024-pair tiles (<= 192 k), 2048-pair (<= 512 k), counted tiles of 4096
# pairs (one tile per histogram workgroup) and of 3072 pairs (four tiles per histogram workgroup), a ragged last tile each
@pytest.mark.parametrize("n", [1, 63, 1025, 150_001, 500_003, 1_200_007, 4_300_001])
@pytest.mark.parametrize("kind,bits", [("uniform", 32), ("five", 16), ("one", 32), ("sorted", 13), ("depth", 32)])
def test_sort_is_stable_and_matches_torch(native_lib, n, kind, bits):
    gen = torch.Generator().manual_seed(n * 31 + bits)
    keys = _keys(kind, n, bits, gen).to(DEV)
    k32 = keys.to(torch.int32) if bits < 32 else (keys - ((keys >> 31) << 32)).to(torch.int32)     # same bit pattern
    got_k, got_v = _sort(native_lib, k32, bits)
    ref_k, ref_v = torch.sort(keys, stable=True)
    assert torch.equal(got_k.long() & 0xFFFFFFFF, ref_k)
    assert torch.equal(got_v.long(), ref_v)


@pytest.mark.parametrize("n", [150_001, 500_003, 1_200_007, 4_300_001])
def test_atomic_ranking_equals_ballot_ranking(native_lib, n):
    gen = torch.Generator().manual_seed(n)
    keys = _keys("five", n, 16, gen).to(DEV).to(torch.int32)
    a = _sort(native_lib, keys, 16)
    native_lib.mgs_debug_set_option(b"radix_ballot_rank", 1)
    try:
        b = _sort(native_lib, keys, 16)
    finally:
        native_lib.mgs_debug_set_option(b"radix_ballot_rank", 0)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize("n", [1, 63, 1025, 150_001, 196_700, 500_003])
@pytest.mark.parametrize("kind,bits", [("five", 16), ("depth", 32)])
def test_block_ids_as_tile_ids_on_an_exclusive_device(native_lib, n, kind, bits):
    """Tile ids come from an atomic ticket; under MGS_FLAG_EXCLUSIVE_DEVICE a one-sweep sort of at most 256 tiles takes block
    ids instead (rs_run).  Same stable order, no error word raised (the entry point returns 2 if one is)."""
    gen = torch.Generator().manual_seed(n * 17 + bits)
    keys = _keys(kind, n, bits, gen).to(DEV)
    k32 = keys.to(torch.int32) if bits < 32 else (keys - ((keys >> 31) << 32)).to(torch.int32)
    native_lib.mgs_debug_set_option(b"debug_sort_exclusive", 1)
    try:
        got = _sort(native_lib, k32, bits)
    finally:
        native_lib.mgs_debug_set_option(b"debug_sort_exclusive", 0)
    ref_k, ref_v = torch.sort(keys, stable=True)
    assert torch.equal(got[0].long() & 0xFFFFFFFF, ref_k) and torch.equal(got[1].long(), ref_v)


@pytest.mark.parametrize("n", [600_011, 1_200_007, 4_300_001])
def test_xcd_band_tile_order_changes_nothing(native_lib, n):
    """The counted-tiles passes take one contiguous band of tiles per XCD (workgroup b -> tile (b mod 8) bands + b / 8) instead
    of tile = block id: only who writes which part of the output changes, not one word of it."""
    gen = torch.Generator().manual_seed(n + 3)
    keys = _keys("depth", n, 32, gen).to(DEV)
    k32 = (keys - ((keys >> 31) << 32)).to(torch.int32)
    a = _sort(native_lib, k32, 32)
    native_lib.mgs_debug_set_option(b"radix_xcd_band", 0)
    try:
        b = _sort(native_lib, k32, 32)
    finally:
        native_lib.mgs_debug_set_option(b"radix_xcd_band", 1)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    ref_k, ref_v = torch.sort(keys, stable=True)
    assert torch.equal(a[0].long() & 0xFFFFFFFF, ref_k) and torch.equal(a[1].long(), ref_v)


def test_both_offset_paths_agree_on_a_large_sort(native_lib):
    gen = torch.Generator().manual_seed(5)
    keys = _keys("depth", 2_000_000, 32, gen).to(DEV)
    k32 = (keys - ((keys >> 31) << 32)).to(torch.int32)
    a = _sort(native_lib, k32, 32)
    native_lib.mgs_debug_set_option(b"radix_scanned", 0)
    try:
        b = _sort(native_lib, k32, 32)
    finally:
        native_lib.mgs_debug_set_option(b"radix_scanned", -1)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_depths_beyond_the_narrow_range_take_the_fourth_pass(native_lib):
    """Large maps sort bits(depth) - bits(0.2) in three 9-bit passes, exact up to depth 13 107; beyond, a device flag
    turns the always-launched fourth pass on.  A scene scaled by 4000 (same image up to the near cull, depths 2 000 -
    34 000) must give the tile lists of the plain four-pass sort (the one-sweep path) bit for bit, sorted by
    (tile, depth bits, index)."""
    from monogs_amd.debug import forward_tables
    from monogs_amd.rasterizer import GaussianRasterizationSettings
    from monogs_amd.synthetic import make_scene, scene_settings
    K = 4000.0
    sc = make_scene(700_000, "fr3_office", seed=11, near_fraction=0.0)
    sc = sc._replace(means3D=sc.means3D * K, scales=sc.scales * K, t=sc.t * K)
    st = scene_settings(sc, GaussianRasterizationSettings, device=DEV)
    dev = lambda x: x.to(DEV)  # noqa: E731
    args = dict(colors_precomp=dev(sc.colors), scales=dev(sc.scales.repeat(1, 3)), rotations=dev(sc.rotations))
    t = forward_tables(st, dev(sc.means3D), dev(sc.opacities), **args)
    assert t["status"] == 0 and t["num_rendered"] > 0
    vis = t["radii"] > 0
    depth = t["rec"][:, 11]
    assert float(depth[vis].max()) > 13107.2 > float(depth[vis].min())          # both sides of the narrow range
    pl = t["point_list"].long()
    key = (t["tile_sorted"].long() << 32) | (t["depth_key"].long() & 0xFFFFFFFF)[pl]
    d = key[1:] - key[:-1]
    assert bool((d >= 0).all()) and bool((pl[1:][d == 0] > pl[:-1][d == 0]).all())
    native_lib.mgs_debug_set_option(b"radix_scanned", 0)
    try:
        t2 = forward_tables(st, dev(sc.means3D), dev(sc.opacities), **args)
    finally:
        native_lib.mgs_debug_set_option(b"radix_scanned", -1)
    assert torch.equal(t["point_list"], t2["point_list"]) and torch.equal(t["ranges"], t2["ranges"])
    assert torch.equal(t["perm"], t2["perm"]) and torch.equal(t["color"], t2["color"])


def test_eight_sorts_on_eight_streams_at_once(native_lib):
    """A mapping window sorts its keyframes at the same time, a stream each.  A tile of a small sort waits (bounded) for the
    digit counts of EARLIER tiles only, so it makes progress whatever else occupies the chip: eight threads sort 415 k
    five-key pairs each (203 tiles: 8 x 203 workgroups in flight, more than fit at once) and every result must be the
    stable order, with no timeout raised."""
    import threading
    from monogs_amd._lib import check
    n, bits = 415_000, 16
    gen = torch.Generator().manual_seed(99)
    jobs = []
    for i in range(8):
        keys = _keys("five" if i % 2 == 0 else "uniform", n, bits, gen).to(DEV)
        k = keys.to(torch.int32).clone()
        v = torch.arange(n, device=DEV, dtype=torch.int32)
        jobs.append(dict(ref=torch.sort(keys, stable=True), k=k, v=v, ka=torch.empty_like(k), va=torch.empty_like(v),
                         temp=torch.empty(native_lib.mgs_debug_sort_temp_bytes(n, bits), dtype=torch.uint8, device=DEV),
                         stream=torch.cuda.Stream(), err=None))
    torch.cuda.synchronize()

    def run(j):
        try:
            for _ in range(4):            # (sorting sorted data again is still a full sort)
                check(native_lib.mgs_debug_sort_pairs(j["k"].data_ptr(), j["v"].data_ptr(), j["ka"].data_ptr(), j["va"].data_ptr(),
                                                      n, bits, j["temp"].data_ptr(), j["stream"].cuda_stream), "mgs_debug_sort_pairs")
        except Exception as e:            # noqa: BLE001
            j["err"] = e

    threads = [threading.Thread(target=run, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    for j in jobs:
        assert j["err"] is None, j["err"]
        assert torch.equal(j["k"].long(), j["ref"][0]) and torch.equal(j["v"].long(), j["ref"][1])
