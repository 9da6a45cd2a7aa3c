"""The tile-pair schedule of the wide crossprod kernels, restated in Python and checked without a GPU.

csrc/crossprod.hip, panels_body at 24 / 32 column tiles (SPLIT = 3 / 4 workgroups share a range of row panels, 8
wavefronts each, one tile row per wavefront): with RT = ceil(ncol / 16) real tiles the launcher picks the instantiation
NAX = RT / 2 + 1; a workgroup ("part") owns NWR = ceil(RT / SPLIT) tile rows from t0 = part * NWR, tile row I meets the
tiles (I + s) mod RT for s < NAX (the last s only for rows below RT / 2 when RT is even), a workgroup densifies the local
tiles 0 .. NWR + RT / 2 - 1 (local t = tile (t0 + t) mod RT), and xp_parts_of_tile says which workgroups a tile's entries
concern (the has[] bits).  What must hold for the reference's t(A) %*% A (RcppSparse.h:159-194: every column pair once):
every unordered tile pair is computed exactly once, every B tile stands inside what its workgroup densified (and no tile
stands there twice), and the has[] bits name exactly the workgroups that densify a tile."""
import pytest


def schedule(nt, split, rt):
    nw = nt // split
    nax = rt // 2 + 1
    nwr = (rt + split - 1) // split
    need = nwr + rt // 2
    wl = nw + nt // 2
    pairs, windows = [], []
    for part in range(split):
        t0 = part * nwr
        used = set()
        for wave in range(nw):
            trow = t0 + wave
            live = wave < nwr and trow < rt
            for s in range(nax):
                has_pair = s < nax - 1 or rt % 2 == 1 or trow < rt // 2
                local_b = wave + s
                assert local_b < wl                      # (the LDS read of a wavefront without a tile row stays inside the buffer too)
                if not (live and has_pair):
                    continue
                tb = (trow + s) % rt
                assert local_b < need, (rt, part, wave, s)
                assert (t0 + local_b) % rt == tb         # the local tile holds the B tile's columns
                used.add(wave)
                used.add(local_b)
                pairs.append((min(trow, tb), max(trow, tb)))
        windows.append((t0, need, used))
    return nax, nwr, need, wl, pairs, windows


def parts_of_tile(tile, rt, split):
    nwr = (rt + split - 1) // split
    need = nwr + rt // 2
    bits = 0
    for h in range(split):
        d = (tile - h * nwr) % rt
        if d < need:
            bits |= 1 << h
    return bits


@pytest.mark.parametrize("nt,split", [(24, 3), (32, 4)])
def test_every_tile_pair_once_at_every_real_tile_count(nt, split):
    for rt in range(nt - 7, nt + 1):
        nax, nwr, need, wl, pairs, windows = schedule(nt, split, rt)
        assert nt // 2 - 3 <= nax <= nt // 2 + 1         # the five instantiations per tile count the launcher has
        assert need <= wl and need <= rt                  # fits the LDS layout; no tile densified twice by one workgroup
        want = sorted((a, b) for a in range(rt) for b in range(a, rt))
        assert sorted(pairs) == want, (nt, rt)
        for part, (t0, n, used) in enumerate(windows):
            live_rows = max(0, min(nwr, rt - t0))
            if live_rows:
                assert max(used) < n
            for tile in range(rt):
                local = (tile - t0) % rt
                assert bool(parts_of_tile(tile, rt, split) >> part & 1) == (local < n), (rt, part, tile)


def test_full_width_is_the_schedule_of_round_4():
    """RT = NT: eight tile rows per workgroup, NT / 2 + 1 pairs for the rows below NT / 2 and one fewer for the others, all
    of the NT / SPLIT + NT / 2 local tiles densified."""
    for nt, split in ((24, 3), (32, 4)):
        nax, nwr, need, wl, pairs, _ = schedule(nt, split, nt)
        assert (nax, nwr, need) == (nt // 2 + 1, nt // split, wl)
        assert len(pairs) == nt * (nt + 1) // 2
