"""Model check of csrc/xwg.h's cross-workgroup reduction protocol (E form with its abandon / sweep fallback, and the L form):
the workgroups of one output tile are run as coroutines that yield at every memory operation, a random scheduler picks who
moves next, a workgroup may be held back arbitrarily long (not yet dispatched) and an owner's poll may run out at any time.
Checked for every schedule: every share of the tile is combined exactly once, by someone who could see every slice's partial
of it; nobody reads a partial that was not published before the reader learnt of it through an atomic; the state words are
zero when the launch ends; nothing waits for a workgroup that has not arrived.  The protocol below is xwg.h's, transcribed
(publish -> drain -> arrive; poll / claim / abandon; last arriver: own share, sweep, reset).  The hardware run of the same
paths: tests/test_qgemm_gpu.py::test_splitk_seam_under_load and profiles/r04/xwg_abandon_path_forced.json."""
import random

import pytest


class Tile:
    def __init__(self, nsl):
        self.nsl = nsl
        self.w0 = 0                      # arrivals
        self.w1 = 0                      # bit q: share q claimed, bit 16 + q: share q abandoned
        self.published = set()           # (slice, share) partials visible in memory
        self.combined = []               # (share, by_slice)
        self.errors = []

    def combine(self, share, by, own_in_regs):
        for s in range(self.nsl):
            if s == by and own_in_regs:
                continue                 # the owner's own partial lives in its registers
            if (s, share) not in self.published:
                self.errors.append(f"slice {by} combines share {share} without slice {s}'s partial")
        self.combined.append((share, by))


def e_form(tile, me, poll_limit):
    """One workgroup (slice `me`) of the E form; yields before every memory operation."""
    nsl = tile.nsl
    for q in range(nsl):                                 # publish the OTHER shares (write-through), then drain
        if q != me:
            yield
            tile.published.add((me, q))
    yield
    before = tile.w0                                     # arrive (returning atomic add)
    tile.w0 += 1
    if before == nsl - 1:                                # last arriver: own share, then the abandoned ones, then reset
        yield
        tile.combine(me, me, True)
        want = ((1 << nsl) - 1) & ~(1 << me)
        while True:                                      # sweep: every other share claimed or abandoned (their owners have arrived)
            yield
            w1 = tile.w1
            if ((w1 | (w1 >> 16)) & want) == want:
                break
        for q in range(nsl):
            if (w1 >> 16) >> q & 1 and q != me:
                yield
                tile.combine(q, me, False)
        yield
        tile.w0 = 0
        tile.w1 = 0
        return
    polls = 0
    while True:                                          # owner: bounded poll of the arrival count
        yield
        if tile.w0 >= nsl:
            ok = True
            break
        polls += 1
        if polls >= poll_limit:
            ok = False
            break
    if ok:
        yield
        tile.w1 |= 1 << me                               # claim (fire and forget)
        yield
        tile.combine(me, me, True)
    else:
        yield
        tile.published.add((me, me))                     # publish the own share too, drain, mark it abandoned
        yield
        tile.w1 |= 0x10000 << me


def l_form(tile, me):
    for q in range(tile.nsl):                            # one "share": the whole tile; publish it
        pass
    yield
    tile.published.add((me, 0))
    yield
    before = tile.w0
    tile.w0 += 1
    if before == tile.nsl - 1:
        yield
        for s in range(tile.nsl):
            if (s, 0) not in tile.published:
                tile.errors.append(f"last arriver {me} reads slice {s} before it was published")
        tile.combined.append((0, me))
        yield
        tile.w0 = 0
        tile.w1 = 0


def run(nsl, seed, form):
    rng = random.Random(seed)
    tile = Tile(nsl)
    poll_limits = [rng.choice([1, 2, 5, 50, 10 ** 6]) for _ in range(nsl)]
    gens = {s: (e_form(tile, s, poll_limits[s]) if form == "E" else l_form(tile, s)) for s in range(nsl)}
    start_after = {s: rng.choice([0, 0, 0, 5, 40, 400]) for s in range(nsl)}      # dispatch delay in scheduler ticks
    tick = 0
    while gens:
        tick += 1
        assert tick < 10 ** 6, "the protocol did not terminate"
        ready = [s for s in gens if start_after[s] <= tick]
        if not ready:
            continue
        s = rng.choice(ready)
        try:
            next(gens[s])
        except StopIteration:
            del gens[s]
    return tile


@pytest.mark.parametrize("nsl", [2, 4])
def test_e_form_every_share_exactly_once_under_any_schedule(nsl):
    took_fallback = 0
    for seed in range(3000):
        tile = run(nsl, seed, "E")
        assert not tile.errors, (seed, tile.errors)
        assert sorted(q for q, _ in tile.combined) == list(range(nsl)), (seed, tile.combined)
        assert tile.w0 == 0 and tile.w1 == 0, (seed, tile.w0, tile.w1)
        took_fallback += any(by != q for q, by in tile.combined)
    assert took_fallback > 100                            # the abandon / sweep path is really exercised


@pytest.mark.parametrize("nsl", [2, 3, 8, 16])
def test_l_form_last_arriver_sees_every_partial(nsl):
    for seed in range(1000):
        tile = run(nsl, seed, "L")
        assert not tile.errors and len(tile.combined) == 1 and tile.w0 == 0, (seed, tile.errors, tile.combined)


@pytest.mark.parametrize("nr,nc,nw", [(8, 2, 8), (4, 2, 8), (2, 2, 8), (1, 2, 8), (4, 2, 12)])
def test_xwg_seam_fragment_shares_and_slab_addresses(nr, nc, nw):
    """csrc/xwg.h `xwg_seam<NR, NC, NW>` (round 5: the split-K kernel's epilogue as a function, used by qgemm_block3.h): which fragment rows
    a slice keeps / publishes, and where they live in the slabs.  E form (2 or 4 slices dividing NR): share q = fragment rows q, q + nsl,
    ...; slice s keeps share s and publishes the others at [slice][tile][wave][row][column tile][lane] x 16 B.  Checked: every fragment
    of every wave is owned by exactly one slice, the owner reads exactly the other slices' copies of ITS fragments from the addresses they
    were written to, no two (slice, tile, wave, fragment, lane) share bytes, and everything stays inside nsl x ntiles x TILE_SLAB."""
    tile_slab = nw * nr * nc * 1024
    ntiles = 5

    def off(slc, tile, wave, i, t, lane):
        return slc * (ntiles * tile_slab) + tile * tile_slab + wave * (nr * nc * 1024) + (i * nc + t) * 1024 + lane * 16

    for nsl in (2, 4, 3, 8):
        e_form = nsl in (2, 4) and nr % nsl == 0
        tile = 3
        written = {}
        for s in range(nsl):
            for wave in range(nw):
                for i in range(nr):
                    if e_form and i % nsl == s:
                        continue                                   # the owner's share stays in registers
                    for t in range(nc):
                        for lane in (0, 17, 63):
                            o = off(s, tile, wave, i, t, lane)
                            assert 0 <= o and o + 16 <= nsl * ntiles * tile_slab
                            assert o not in written
                            written[o] = (s, wave, i, t, lane)
        owners = {}
        for s in range(nsl):                                       # what each slice combines (E: its share; L: the last arriver, everything)
            for wave in range(nw):
                rows = [i for i in range(nr) if i % nsl == s] if e_form else (list(range(nr)) if s == nsl - 1 else [])
                for i in rows:
                    owners.setdefault((wave, i), []).append(s)
                    for t in range(nc):
                        for other in range(nsl):
                            if other == s and e_form:
                                continue
                            assert written[off(other, tile, wave, i, t, 17)] == (other, wave, i, t, 17)
        assert all(len(v) == 1 for v in owners.values()) and len(owners) == nw * nr

